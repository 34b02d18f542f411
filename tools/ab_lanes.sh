run() { echo "== $*"; env "$@" timeout 300 python bench.py --steps 128 --warmup 64 --no-cpu-baseline 2>&1 | python -c "
import sys,json
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); print(d['value'], d['ms_per_step'], d['roofline']['stage_ms_per_step'])
"; }
run FSPT_WF_LANES=1
run FSPT_WF_LANES=2
for t in 3 4 5; do for l in 1 2; do run FSPT_WF_LANES=2 FSPT_TRACE_BPC=$t FSPT_LOGIC_BPC=$l; done; done
run FSPT_WF_LANES=2 FSPT_TRACE_BPC=8 FSPT_LOGIC_BPC=1
run FSPT_WF_LANES=2 FSPT_TRACE_BPC=4 FSPT_LOGIC_BPC=4
