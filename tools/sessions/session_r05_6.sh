O=gpurun_out/r05; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/gpu_suite_2.log 2>&1; tail -4 $O/gpu_suite_2.log
for i in 1 2; do timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_driver_form_$i.json.log 2>/dev/null; python3 -c "
import json
for l in open('$O/bench_driver_form_$i.json.log'):
    if l.startswith('{'):
        d=json.loads(l); print('driver form', d['value'], d['ms_per_step'], d['rep_ms_per_step'], d['stage_events']['region_ms_per_step'], {c:v['ms_per_step'] for c,v in d['roofline']['kernels'].items()}, {k:(v['value'], v['parity_check']['equal']) for k,v in d.get('extra_configs',{}).items()}, d['parity_check']['equal'])"; done
timeout 900 python bench.py --no-extra-configs > $O/bench_default_128.json.log 2>/dev/null; python3 -c "
import json
for l in open('$O/bench_default_128.json.log'):
    if l.startswith('{'):
        d=json.loads(l); print('128 steps', d['value'], d['ms_per_step'], {c:v['ms_per_step'] for c,v in d['roofline']['kernels'].items()})"
tools/ab_args.sh "--steps 1 --warmup 3 --reps 9" "--stage-events all" "--stage-events last" > $O/bench_tick1.log 2>&1; cat $O/bench_tick1.log
tools/ab_args.sh "--steps 20 --warmup 5" "--pipeline stream" "--textured" "--width 3840 --height 2160" > $O/bench_other_configs.log 2>&1; cat $O/bench_other_configs.log
python tools/write_bench_scene.py /tmp/benchscene > /dev/null && node fspt_amd/js/bench.js --scene /tmp/benchscene/scene/bench.json --focal-depth 2 --aperture 0.02 --steps 20 --warmup 5 > $O/bench_node_host.json.log 2>&1; cat $O/bench_node_host.json.log | cut -c1-400
FSPT_FUZZ_SEEDS=400 timeout 2400 python -m pytest tests/test_parity_gpu.py -m gpu -q -k "fuzz_random_scenes" -p no:cacheprovider > $O/fuzz_soak_400_seeds.log 2>&1; tail -3 $O/fuzz_soak_400_seeds.log
