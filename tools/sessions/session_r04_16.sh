# the round after which the tail kernel takes over: forced values against the adaptive rule, 1 M-triangle scene and C2
O=gpurun_out/r04; mkdir -p $O
{
for rep in 1 2; do
for cfg in "--steps 20 --warmup 5 --config c3" "--steps 20 --warmup 5"; do
for tl in -1 3 4 5 6 7 0; do
  echo -n "== $cfg tail $tl: "
  timeout 900 python3 bench.py $cfg --tail $tl --no-cpu-baseline --no-l1-microbench 2>/dev/null | python3 -c "
import sys,json
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); k=d['roofline'].get('kernels',{})
        print(d['value'], d['ms_per_step'], {c:v['ms_per_step'] for c,v in k.items()}, {c:v['launches'] for c,v in k.items()}, 'parity', (d.get('parity_check') or {}).get('equal'))
"
done; done; done
} > $O/s16_tail_round.log 2>&1
cat $O/s16_tail_round.log
