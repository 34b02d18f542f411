O=gpurun_out/r05; mkdir -p $O
tools/ab_run.sh "--steps 20 --warmup 5" r5b7 r5b6t16 r5b6t31 r5b6t63 > $O/ab_trace_lds_top_c2_20.log 2>&1; cat $O/ab_trace_lds_top_c2_20.log
tools/ab_run.sh "--steps 128 --warmup 128" r5b7 r5b6t31 r5b6t63 > $O/ab_trace_lds_top_c2_128.log 2>&1; cat $O/ab_trace_lds_top_c2_128.log
tools/ab_run.sh "--config c3 --steps 20 --warmup 5" r5b7 r5b6t31 r5b6t63 > $O/ab_trace_lds_top_c3.log 2>&1; cat $O/ab_trace_lds_top_c3.log
