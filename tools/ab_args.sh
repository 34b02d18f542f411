#!/bin/bash
# usage: tools/ab_args.sh "<common bench.py args>" "<variant args 1>" "<variant args 2>" ...
# One library, variants differ in bench.py arguments; same box, interleaved, two passes.
common="$1"; shift
for rep in 1 2; do
for v in "$@"; do
echo -n "== [$v] (rep $rep): "
timeout 900 python3 bench.py $common $v --no-cpu-baseline --no-extra-configs --no-l1-microbench 2>/dev/null | python3 -c "
import sys,json
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); k=d['roofline'].get('kernels',{})
        print(d['value'], d['ms_per_step'], {c:v['ms_per_step'] for c,v in k.items()}, 'parity', (d.get('parity_check') or {}).get('equal'))
"
done; done
