# the tail hand-over round again, now that the last trace launch suspends: single tick and the 1 M-triangle scene
O=gpurun_out/r04; mkdir -p $O
{
for cfg in "--steps 1 --warmup 3" "--steps 20 --warmup 5 --config c3"; do
for tl in -1 1 2 3 4 5 6; do
  echo -n "== $cfg tail $tl: "
  timeout 900 python3 bench.py $cfg --tail $tl --no-cpu-baseline --no-l1-microbench 2>/dev/null | python3 -c "
import sys,json
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); k=d['roofline'].get('kernels',{})
        print(d['value'], d['ms_per_step'], {c:v['ms_per_step'] for c,v in k.items()}, {c:v['launches'] for c,v in k.items()}, 'parity', (d.get('parity_check') or {}).get('equal'))
"
done; done
} > $O/s35_tail_round_after.log 2>&1
cat $O/s35_tail_round_after.log
