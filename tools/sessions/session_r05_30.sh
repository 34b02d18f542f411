O=gpurun_out/r05; mkdir -p $O
V=("--trace-budget 24" "--trace-budget 16" "--trace-budget 12" "--trace-budget 8")
tools/ab_args.sh "--steps 20 --warmup 5" "${V[@]}" > $O/scan_trace_budget_low_c2_20.log 2>&1; cat $O/scan_trace_budget_low_c2_20.log
tools/ab_args.sh "--config c3 --steps 20 --warmup 5" "${V[@]}" > $O/scan_trace_budget_low_c3.log 2>&1; cat $O/scan_trace_budget_low_c3.log
