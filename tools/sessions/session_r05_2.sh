O=gpurun_out/r05; mkdir -p $O
tools/ab_run.sh "--steps 20 --warmup 5" r5base r5carry r5clears r5both > $O/ab_fixed_costs_c2_20.log 2>&1; cat $O/ab_fixed_costs_c2_20.log
tools/ab_run.sh "--steps 1 --warmup 3 --reps 9" r5base r5both > $O/ab_fixed_costs_tick1.log 2>&1; cat $O/ab_fixed_costs_tick1.log
tools/ab_run.sh "--config c3 --steps 20 --warmup 5" r5base r5both > $O/ab_fixed_costs_c3.log 2>&1; cat $O/ab_fixed_costs_c3.log
timeout 2400 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/gpu_suite_1.log 2>&1; tail -5 $O/gpu_suite_1.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_default_extra.json.log 2> $O/bench_default_extra.err; tail -c 3000 $O/bench_default_extra.json.log; tail -3 $O/bench_default_extra.err
