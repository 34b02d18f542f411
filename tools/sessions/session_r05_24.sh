O=gpurun_out/r05; mkdir -p $O
tools/ab_run.sh "--steps 20 --warmup 5" r5t_base r5t_grid6 r5t_grid7 r5t_over1 > $O/ab_grid_sizes_c2_20.log 2>&1; cat $O/ab_grid_sizes_c2_20.log
tools/ab_run.sh "--config c3 --steps 20 --warmup 5" r5t_base r5t_grid6 r5t_grid7 r5t_over1 > $O/ab_grid_sizes_c3.log 2>&1; cat $O/ab_grid_sizes_c3.log
tools/ab_run.sh "--steps 1 --warmup 3 --reps 9" r5t_base r5t_grid6 r5t_over1 > $O/ab_grid_sizes_tick1.log 2>&1; cat $O/ab_grid_sizes_tick1.log
