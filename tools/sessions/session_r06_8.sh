# usage (GPU box): bash tools/sessions/session_r06_8.sh
# the scheduling constants re-scanned on the round-6 kernels (their optima were found with a trace kernel that reloaded a
# spilled address after every leaf visit): one box, interleaved, two passes
O=gpurun_out/r06; mkdir -p $O
bash tools/ab_run.sh "--steps 20 --warmup 5" t_base t_im8 t_im24 t_ch64 t_ch256 t_tc06 t_tc13 t_lds63 t_ts16 t_ts64 t_lu4 t_fine4 > $O/scan_constants_c2_20.log 2>&1; cat $O/scan_constants_c2_20.log
bash tools/ab_run.sh "--steps 20 --warmup 5 --config c3" t_base t_im8 t_im24 t_ch64 t_ch256 t_tc06 t_tc13 t_lds63 t_ts16 t_ts64 t_lu4 t_fine4 > $O/scan_constants_c3.log 2>&1; cat $O/scan_constants_c3.log
