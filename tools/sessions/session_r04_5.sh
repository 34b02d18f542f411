O=gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/s5_pytest.log 2>&1; tail -3 $O/s5_pytest.log
tools/ab_run.sh "--steps 1 --warmup 3" tp t1 t1s16 t1s64 > $O/ab_tail_lane_tick1.log 2>&1; cat $O/ab_tail_lane_tick1.log
tools/ab_run.sh "--steps 20 --warmup 5 --config c3" tp t1 t1s16 t1s64 > $O/ab_tail_lane_c3.log 2>&1; cat $O/ab_tail_lane_c3.log
tools/ab_run.sh "--steps 20 --warmup 5" tp t1 t1s16 t1s64 > $O/ab_tail_lane.log 2>&1; cat $O/ab_tail_lane.log
echo "== t1prof tick1"; FSPT_LIB=$PWD/ab_libs/t1prof.so timeout 300 python3 bench.py --steps 1 --warmup 3 --reps 1 --no-cpu-baseline --no-l1-microbench --no-parity-check 2>&1 | grep -E "^tailprof" | tail -8
