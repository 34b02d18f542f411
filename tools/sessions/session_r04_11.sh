O=gpurun_out/r04; mkdir -p $O
for i in 1 2 3; do
for a in "--steps 20 --warmup 5 --config c3" "--steps 20 --warmup 5"; do
  echo "== bench.py $a"
  timeout 900 python3 bench.py $a --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); k=d['roofline'].get('kernels',{})
        print(d['value'], d['ms_per_step'], d['rep_ms_per_step'], {c:v['ms_per_step'] for c,v in k.items()}, d['config'].get('primary_form'))
"
done; done > $O/s11_tuner_repeat.log 2>&1
cat $O/s11_tuner_repeat.log
