O=gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/s10_pytest.log 2>&1; tail -12 $O/s10_pytest.log
for a in "--steps 20 --warmup 5" "--steps 20 --warmup 5 --config c3" "--steps 1 --warmup 3" "--steps 128 --warmup 128" "--steps 20 --warmup 5 --config c3 --primary-form 1" "--steps 20 --warmup 5 --config c3 --primary-form 2"; do
  echo "== bench.py $a"
  timeout 900 python3 bench.py $a --no-cpu-baseline --no-l1-microbench 2>/dev/null | python3 -c "
import sys,json
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); k=d['roofline'].get('kernels',{})
        print(d['value'], 'Msamples/s', d['ms_per_step'], 'ms/step', d['rep_ms_per_step'], {c:v['ms_per_step'] for c,v in k.items()}, d['config'].get('primary_form'), 'parity', (d.get('parity_check') or {}).get('equal'))
"
done > $O/s10_tuner.log 2>&1
cat $O/s10_tuner.log
