O=gpurun_out/r05; mkdir -p $O
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 2400 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/gpu_suite_3.log 2>&1; grep -E "passed|failed" $O/gpu_suite_3.log | tail -2
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_bench.json.log 2> $O/driver_bench.err; python3 -c "
import json
for l in open('$O/driver_bench.json.log'):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print(d['value'], d['ms_per_step'], d['rep_ms_per_step'], 'frac', r['frac'], 'hbm_counter', r['hbm_counter'] and r['hbm_counter']['frac'], 'primary', r['kernels']['primary'].get('frac'), 'blend c3', d['extra_configs']['c3']['roofline'].get('blended_with_l2_hit_rate'), {k:(v['value'],v['parity_check']['equal']) for k,v in d['extra_configs'].items()}, d['parity_check']['equal'], d['cpu_baseline']['value'])"
