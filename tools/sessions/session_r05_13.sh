O=gpurun_out/r05; mkdir -p $O
tools/ab_run.sh "--width 480 --height 270 --steps 1 --warmup 3 --reps 15" r5pairs32 r5pairsauto > $O/ab_tail_pairs_auto_480x270_tick1.log 2>&1; cat $O/ab_tail_pairs_auto_480x270_tick1.log
tools/ab_run.sh "--width 960 --height 540 --steps 1 --warmup 3 --reps 15" r5pairs32 r5pairsauto > $O/ab_tail_pairs_auto_960x540_tick1.log 2>&1; cat $O/ab_tail_pairs_auto_960x540_tick1.log
tools/ab_run.sh "--width 480 --height 270 --steps 20 --warmup 5" r5pairs32 r5pairsauto > $O/ab_tail_pairs_auto_480x270_20.log 2>&1; cat $O/ab_tail_pairs_auto_480x270_20.log
tools/ab_run.sh "--steps 20 --warmup 5" r5pairs32 r5pairsauto > $O/ab_tail_pairs_auto_c2_20.log 2>&1; cat $O/ab_tail_pairs_auto_c2_20.log
rm -rf $O/final_* $O/c3_* $O/c5_*
timeout 1800 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/gpu_suite_final.log 2>&1; grep -E "passed|failed" $O/gpu_suite_final.log | tail -1
bash tools/sessions/session_r05_prof.sh
