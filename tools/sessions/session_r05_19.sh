O=gpurun_out/r05; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/gpu_suite_head.log 2>&1; tail -2 $O/gpu_suite_head.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/head_bench_steps20.json.log 2>$O/head_bench_steps20.err; python3 -c "
import json
for l in open('$O/head_bench_steps20.json.log'):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print(d['value'], r['frac'], 'traffic', r['traffic'], r.get('traffic_source'), {k:v['value'] for k,v in d['extra_configs'].items()})
"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
