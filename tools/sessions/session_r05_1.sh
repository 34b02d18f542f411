O=gpurun_out/r05; mkdir -p $O
timeout 1500 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "two_level or deep_chain or fuzz or not_unions or intersect" -p no:cacheprovider > $O/tests_two_level.log 2>&1; tail -5 $O/tests_two_level.log
V=("--node-form 0,0,0" "--node-form 0,0,1" "--node-form 0,1,1" "--node-form 1,0,1" "--node-form 0,-1,1,300000" "--node-form 0,-1,1,1500000")
tools/ab_args.sh "--steps 20 --warmup 5" "${V[@]}" > $O/ab_two_level_c2_20.log 2>&1; cat $O/ab_two_level_c2_20.log
tools/ab_args.sh "--steps 1 --warmup 3 --reps 9" "${V[@]}" > $O/ab_two_level_c2_tick1.log 2>&1; cat $O/ab_two_level_c2_tick1.log
tools/ab_args.sh "--config c3 --steps 20 --warmup 5" "--node-form 0,0,0" "--node-form 0,0,1" "--node-form 0,1,1" "--node-form 1,1,1" "--node-form 1,0,1" > $O/ab_two_level_c3_20.log 2>&1; cat $O/ab_two_level_c3_20.log
