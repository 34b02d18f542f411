# one vs two lanes over the step count (same box)
for k in 16 32 64 128 256; do
for pl in wavefront wavefront2; do
echo -n "$pl steps=$k : "
timeout 300 python bench.py --steps $k --warmup $k --pipeline $pl --no-cpu-baseline 2>&1 | python -c "
import sys,json
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); print(d['value'], d['ms_per_step'], d['roofline']['stage_ms_per_step'])
"
done; done
