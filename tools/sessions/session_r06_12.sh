# usage (GPU box): bash tools/sessions/session_r06_12.sh
# a second look at constants that matter on the 1 M-triangle scene (latency-bound launches): primary refill slice, tail slice
# 8 / 12 / 16, interior-min 28, 32 pool heads, hand-over coefficient 1.1, suspension budget 16 / 32, tail at 5 waves, lag cap 2
O=gpurun_out/r06; mkdir -p $O
bash tools/ab_run.sh "--steps 20 --warmup 5 --config c3" g_base g_ps4 g_ps16 g_tsl8 g_tsl12 g_im28 g_heads32 g_tc11 g_bud16 g_bud32 g_tw5 g_lag2 > $O/scan_constants3_c3.log 2>&1; cat $O/scan_constants3_c3.log
bash tools/ab_run.sh "--steps 20 --warmup 5" g_base g_im28 g_heads32 g_tc11 g_bud16 g_bud32 g_lag2 > $O/scan_constants3_c2_20.log 2>&1; cat $O/scan_constants3_c2_20.log
