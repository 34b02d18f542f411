# usage (GPU box): bash tools/sessions/session_r06_6.sh
# cooperative leaf visits in the tail kernel's lane pairs (trace_slice_pair, WF_TAIL_COOP): the tail / stream / fuzz tests on
# the new kernel, then the A/B: single tick, C2 20 ticks, 1 M triangles, quarter-resolution preview
O=gpurun_out/r06; mkdir -p $O
FSPT_LIB=$PWD/ab_libs/coop1.so timeout 1500 python3 -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "tail or stream or fuzz or refract or render or suspended or chain or baseline or million" > $O/gpu_coop_tests.log 2>&1; tail -2 $O/gpu_coop_tests.log
bash tools/ab_run.sh "--steps 1 --warmup 3 --reps 15" coop0 coop1 > $O/ab_tail_coop_tick1.log 2>&1; cat $O/ab_tail_coop_tick1.log
bash tools/ab_run.sh "--steps 20 --warmup 5" coop0 coop1 > $O/ab_tail_coop_c2_20.log 2>&1; cat $O/ab_tail_coop_c2_20.log
bash tools/ab_run.sh "--steps 20 --warmup 5 --config c3" coop0 coop1 > $O/ab_tail_coop_c3.log 2>&1; cat $O/ab_tail_coop_c3.log
bash tools/ab_run.sh "--steps 1 --warmup 3 --reps 15 --width 480 --height 270" coop0 coop1 > $O/ab_tail_coop_480x270_tick1.log 2>&1; cat $O/ab_tail_coop_480x270_tick1.log
