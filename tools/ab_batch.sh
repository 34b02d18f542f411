for b in 64 128 256; do
echo -n "batch $b : "
FSPT_LIB=$PWD/ab_libs/b256.so timeout 300 python bench.py --steps 256 --warmup $b --batch $b --no-cpu-baseline 2>&1 | python -c "
import sys,json
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); print(d['value'], d['ms_per_step'], d['roofline']['stage_ms_per_step'])
"
done
