# the tuner with split batches: the forms test, then the choice and the regions of C3 / C2 three times over
O=gpurun_out/r04; mkdir -p $O
{
timeout 900 python3 -m pytest tests/test_parity_gpu.py -q -x -k "primary_launch_forms or two_call or deferred or memory_limit or fuzz" 2>&1 | tail -3
for i in 1 2 3; do
for a in "--steps 20 --warmup 5 --config c3" "--steps 20 --warmup 5" "--steps 128 --warmup 128" "--steps 1 --warmup 3"; do
  echo "== bench.py $a"
  timeout 900 python3 bench.py $a --no-cpu-baseline --no-l1-microbench 2>/dev/null | python3 -c "
import sys,json
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); k=d['roofline'].get('kernels',{})
        print(d['value'], d['ms_per_step'], d['rep_ms_per_step'], {c:v['ms_per_step'] for c,v in k.items()}, d['config'].get('primary_form'), (d.get('parity_check') or {}).get('equal'))
"
done; done
} > $O/s18_tuner_split.log 2>&1
cat $O/s18_tuner_split.log
