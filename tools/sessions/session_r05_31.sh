O=gpurun_out/r05; mkdir -p $O
tools/ab_run.sh "--steps 20 --warmup 5" r5g8 r5gres > $O/ab_trace_grid_resident_c2_20.log 2>&1; cat $O/ab_trace_grid_resident_c2_20.log
tools/ab_run.sh "--config c3 --steps 20 --warmup 5" r5g8 r5gres > $O/ab_trace_grid_resident_c3.log 2>&1; cat $O/ab_trace_grid_resident_c3.log
tools/ab_run.sh "--steps 128 --warmup 128" r5g8 r5gres > $O/ab_trace_grid_resident_c2_128.log 2>&1; cat $O/ab_trace_grid_resident_c2_128.log
