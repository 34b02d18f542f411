# usage: tools/ab_smallk.sh name1 name2 ... : bench at 128 steps + small-batch probe per library (same box)
for n in "$@"; do
echo "== $n"
FSPT_LIB=$PWD/ab_libs/$n.so timeout 300 python bench.py --steps 128 --warmup 64 --no-cpu-baseline 2>&1 | python -c "
import sys,json
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); print(d['value'], d['ms_per_step'], d['roofline']['stage_ms_per_step'])
"
FSPT_LIB=$PWD/ab_libs/$n.so timeout 300 python tools/floor_probe.py 1920 1080 1 4 8 16 2>&1 | grep wall
done
