# usage (GPU box): bash tools/sessions/session_r06_10.sh
# the tree with WF_INTERIOR_MIN 24 and the tail slice chosen by scene size (f_base) against round 6's earlier constants
# (f_old), and with the traversal stack one entry shorter (f_spare0: the 1 M-triangle scene - depth 22 - then fits its
# sixth trace block per CU): deep-tree / suspension / fuzz tests on the shorter stack first, then the A/B
O=gpurun_out/r06; mkdir -p $O
FSPT_LIB=$PWD/ab_libs/f_spare0.so timeout 1500 python3 -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "chain or deep or suspended or fuzz or tail or stream or million or intersect or bvh_test or two_level" > $O/gpu_spare0_tests.log 2>&1; tail -2 $O/gpu_spare0_tests.log
bash tools/ab_run.sh "--steps 20 --warmup 5 --config c3" f_old f_base f_spare0 > $O/ab_final_constants_c3.log 2>&1; cat $O/ab_final_constants_c3.log
bash tools/ab_run.sh "--steps 20 --warmup 5" f_old f_base f_spare0 > $O/ab_final_constants_c2_20.log 2>&1; cat $O/ab_final_constants_c2_20.log
bash tools/ab_run.sh "--steps 128 --warmup 128 --reps 3 --config c3" f_old f_base f_spare0 > $O/ab_final_constants_c3_128.log 2>&1; cat $O/ab_final_constants_c3_128.log
bash tools/ab_run.sh "--steps 1 --warmup 3 --reps 15" f_old f_base > $O/ab_final_constants_tick1.log 2>&1; cat $O/ab_final_constants_tick1.log
