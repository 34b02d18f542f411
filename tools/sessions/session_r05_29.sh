O=gpurun_out/r05; mkdir -p $O
V=("--trace-budget 48" "--trace-budget 24" "--trace-budget 96" "--trace-budget 192" "--trace-budget 0")
tools/ab_args.sh "--config c3 --steps 20 --warmup 5" "${V[@]}" > $O/scan_trace_budget_tickets_c3.log 2>&1; cat $O/scan_trace_budget_tickets_c3.log
tools/ab_args.sh "--steps 20 --warmup 5" "${V[@]}" > $O/scan_trace_budget_tickets_c2_20.log 2>&1; cat $O/scan_trace_budget_tickets_c2_20.log
