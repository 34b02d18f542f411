# usage (GPU box): bash tools/sessions/session_r06_4.sh
# the tree after session 3's decisions (wave index scalar: adopted; bins, row-wise resolve, footprint records: not): the GPU
# suite, the driver's bench form with the new extra_configs (textured, tick1) + reference_glsl_baseline, and what the
# suspended traversals' lag costs the tail launch (trace budget 0 = never suspend: no path lags)
O=gpurun_out/r06; mkdir -p $O
timeout 1500 python3 -m pytest tests -q -m gpu > $O/gpu_suite_2.log 2>&1; tail -4 $O/gpu_suite_2.log
( time timeout 900 python3 bench.py --steps 20 --warmup 5 ) > $O/bench_driver_form_1.json.log 2> $O/bench_driver_form_1.time; tail -c 1500 $O/bench_driver_form_1.json.log; cat $O/bench_driver_form_1.time
bash tools/ab_args.sh "--steps 20 --warmup 5" "" "--trace-budget 0" "--trace-budget 96" > $O/scan_budget_c2_20.log 2>&1; cat $O/scan_budget_c2_20.log
bash tools/ab_args.sh "--steps 20 --warmup 5 --config c3" "" "--trace-budget 0" "--trace-budget 96" > $O/scan_budget_c3.log 2>&1; cat $O/scan_budget_c3.log
bash tools/ab_args.sh "--steps 1 --warmup 3 --reps 9" "" "--trace-budget 0" > $O/scan_budget_tick1.log 2>&1; cat $O/scan_budget_tick1.log
