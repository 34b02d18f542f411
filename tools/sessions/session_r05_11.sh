O=gpurun_out/r05; mkdir -p $O
tools/ab_run.sh "--steps 20 --warmup 5 --tail 8" r5p32 r5p8 r5p2 r5p1 > $O/ab_tail_pairs_c2_20_tail8.log 2>&1; cat $O/ab_tail_pairs_c2_20_tail8.log
tools/ab_run.sh "--steps 20 --warmup 5 --tail 6" r5p32 r5p8 r5p2 > $O/ab_tail_pairs_c2_20_tail6.log 2>&1; cat $O/ab_tail_pairs_c2_20_tail6.log
tools/ab_run.sh "--steps 1 --warmup 3 --reps 9 --tail 6" r5p32 r5p8 r5p2 r5p1 > $O/ab_tail_pairs_tick1_tail6.log 2>&1; cat $O/ab_tail_pairs_tick1_tail6.log
tools/ab_run.sh "--config c3 --steps 20 --warmup 5 --tail 8" r5p32 r5p8 r5p2 r5p1 > $O/ab_tail_pairs_c3_tail8.log 2>&1; cat $O/ab_tail_pairs_c3_tail8.log
