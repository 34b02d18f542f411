#!/bin/bash
# usage: tools/ab_run.sh "<bench.py args>" name1 name2 ...   (libraries ab_libs/<name>.so; same box, interleaved, two passes)
args="$1"; shift
for rep in 1 2; do
for n in "$@"; do
echo -n "== $n (rep $rep): "
FSPT_LIB=$PWD/ab_libs/$n.so timeout 600 python3 bench.py $args --no-cpu-baseline --no-extra-configs --no-l1-microbench 2>/dev/null | python3 -c "
import sys,json
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); k=d['roofline'].get('kernels',{})
        print(d['value'], d['ms_per_step'], {c:v['ms_per_step'] for c,v in k.items()}, 'parity', (d.get('parity_check') or {}).get('equal'))
"
done; done
