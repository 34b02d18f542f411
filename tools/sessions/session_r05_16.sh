O=gpurun_out/r05; mkdir -p $O
FSPT_LIB=$PWD/ab_libs/r5pf2.so timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -p no:cacheprovider -k "render_fused or suspended or fuzz or two_call or refractive" > $O/gpu_tail_prefetch_tests.log 2>&1; grep -E "passed|failed|rror" $O/gpu_tail_prefetch_tests.log | tail -2
tools/ab_run.sh "--steps 1 --warmup 3 --reps 9" r5base r5pf1 r5pf2 > $O/ab_tail_prefetch_tick1.log 2>&1; cat $O/ab_tail_prefetch_tick1.log
tools/ab_run.sh "--steps 20 --warmup 5" r5base r5pf1 r5pf2 > $O/ab_tail_prefetch_c2_20.log 2>&1; cat $O/ab_tail_prefetch_c2_20.log
tools/ab_run.sh "--config c3 --steps 20 --warmup 5" r5base r5pf1 r5pf2 > $O/ab_tail_prefetch_c3.log 2>&1; cat $O/ab_tail_prefetch_c3.log
