# usage: tools/ab_libs.sh name1 name2 ...   (libraries ab_libs/<name>.so, same box, interleaved twice)
for rep in 1 2; do
for n in "$@"; do
echo "== $n (rep $rep)"
FSPT_LIB=$PWD/ab_libs/$n.so timeout 300 python bench.py --steps 256 --warmup 128 --no-cpu-baseline 2>&1 | python -c "
import sys,json
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); print(d['value'], d['ms_per_step'], d['roofline']['stage_ms_per_step'])
"
done; done
