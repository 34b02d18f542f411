O=gpurun_out/r05; mkdir -p $O
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -p no:cacheprovider -k "tail_stages or refractive" > $O/gpu_tail_stages_tests.log 2>&1; grep -E "passed|failed|rror" $O/gpu_tail_stages_tests.log | tail -3
V=("--tail-stages 0" "--tail-stages 32" "--tail-stages 32,4" "--tail-stages 32,8" "--tail-stages 16" "--tail-stages 8" "--tail-stages 16,2")
tools/ab_args.sh "--steps 1 --warmup 3 --reps 9" "${V[@]}" > $O/ab_tail_stages_tick1.log 2>&1; cat $O/ab_tail_stages_tick1.log
tools/ab_args.sh "--steps 20 --warmup 5" "${V[@]}" > $O/ab_tail_stages_c2_20.log 2>&1; cat $O/ab_tail_stages_c2_20.log
tools/ab_args.sh "--config c3 --steps 20 --warmup 5" "${V[@]}" > $O/ab_tail_stages_c3.log 2>&1; cat $O/ab_tail_stages_c3.log
