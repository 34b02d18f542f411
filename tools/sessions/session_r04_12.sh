# (experiment of round 4 that was NOT adopted: the sources it ran - primary form 3 / the tail kernel with an LDS copy of the top of the tree - are not in the tree; results: profiles/r04/README.md)
# forms of the primary launch: 1 fused, 2 fused + refill, 3 split (k_wf_camtrace + shading)
O=gpurun_out/r04; mkdir -p $O
{
timeout 600 python3 -m pytest tests/test_parity_gpu.py -q -x -k "primary_launch_forms or bench_configuration or two_call" 2>&1 | tail -3
for rep in 1 2; do
for cfg in "--steps 20 --warmup 5" "--steps 20 --warmup 5 --config c3" "--steps 1 --warmup 3 --batch 1" "--steps 128 --warmup 128"; do
for f in 1 2 3; do
  echo -n "== $cfg form $f: "
  timeout 900 python3 bench.py $cfg --primary-form $f --no-cpu-baseline --no-l1-microbench 2>/dev/null | python3 -c "
import sys,json
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); k=d['roofline'].get('kernels',{})
        print(d['value'], d['ms_per_step'], {c:v['ms_per_step'] for c,v in k.items()}, 'parity', (d.get('parity_check') or {}).get('equal'))
"
done; done; done
} > $O/s12_split.log 2>&1
cat $O/s12_split.log
