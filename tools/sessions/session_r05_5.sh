O=gpurun_out/r05; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "two_level or tail or halved or stream_pool or js_bench or fuzz or deferred or two_call" > $O/gpu_new_tests_3.log 2>&1; tail -4 $O/gpu_new_tests_3.log
tools/ab_run.sh "--steps 20 --warmup 5" r5ownstream r5tstream > $O/ab_target_stream_c2_20.log 2>&1; cat $O/ab_target_stream_c2_20.log
tools/ab_run.sh "--steps 1 --warmup 3 --reps 9" r5ownstream r5tstream > $O/ab_target_stream_tick1.log 2>&1; cat $O/ab_target_stream_tick1.log
V=("--node-form 0,0,0" "--node-form 0,0,2" "--node-form 0,0,1")
tools/ab_args.sh "--steps 20 --warmup 5" "${V[@]}" > $O/ab_tail_adaptive_c2_20.log 2>&1; cat $O/ab_tail_adaptive_c2_20.log
tools/ab_args.sh "--steps 1 --warmup 3 --reps 9" "${V[@]}" > $O/ab_tail_adaptive_tick1.log 2>&1; cat $O/ab_tail_adaptive_tick1.log
tools/ab_args.sh "--config c3 --steps 20 --warmup 5" "${V[@]}" > $O/ab_tail_adaptive_c3.log 2>&1; cat $O/ab_tail_adaptive_c3.log
echo "== stage events off"; for i in 1 2; do FSPT_STAGE_EVENTS=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs --no-l1-microbench 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('events off', d['value'], d['ms_per_step'])"; python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs --no-l1-microbench 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('events on ', d['value'], d['ms_per_step'])"; done > $O/ab_stage_events.log 2>&1; cat $O/ab_stage_events.log
