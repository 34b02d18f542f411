O=gpurun_out/r05; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "tail or fuzz or deep_chain or stream or refractive" > $O/gpu_tail_tests.log 2>&1; grep -E "passed|failed" $O/gpu_tail_tests.log | tail -1
tools/ab_args.sh "--steps 20 --warmup 5" "--tail -1" "--tail 4" "--tail 5" "--tail 6" "--tail 7" "--tail 8" > $O/scan_tail_round_auto_pairs_c2_20.log 2>&1; cat $O/scan_tail_round_auto_pairs_c2_20.log
tools/ab_args.sh "--steps 1 --warmup 3 --reps 9" "--tail -1" "--tail 2" "--tail 3" "--tail 4" "--tail 5" "--tail 6" > $O/scan_tail_round_auto_pairs_tick1.log 2>&1; cat $O/scan_tail_round_auto_pairs_tick1.log
tools/ab_args.sh "--config c3 --steps 20 --warmup 5" "--tail -1" "--tail 5" "--tail 6" "--tail 7" "--tail 8" > $O/scan_tail_round_auto_pairs_c3.log 2>&1; cat $O/scan_tail_round_auto_pairs_c3.log
tools/ab_args.sh "--steps 4 --warmup 4" "--tail -1" "--tail 3" "--tail 4" "--tail 5" "--tail 6" > $O/scan_tail_round_auto_pairs_c2_4.log 2>&1; cat $O/scan_tail_round_auto_pairs_c2_4.log
