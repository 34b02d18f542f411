#!/bin/bash
# Vector-L1 (TA/TCP/TD) counters for the bench kernels and for the gather microbenchmark (calibration).
# usage: tools/pmc_l1.sh <outdir-under-gpurun_out>
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1; shift
mkdir -p $O
cd /tmp
i=0
for set in \
 "TA_TA_BUSY_sum TA_BUSY_avr TD_TD_BUSY_sum GRBM_GUI_ACTIVE" \
 "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum" \
 "TCP_TAGRAM0_REQ_sum TCP_TAGRAM1_REQ_sum TCP_TAGRAM2_REQ_sum TCP_TAGRAM3_REQ_sum" \
 "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
 "TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $O/b$i -- python3 $R/bench.py --warmup 0 --no-cpu-baseline "$@" > $O/b$i.log 2>&1
  timeout 120 rocprofv3 --pmc $set --output-format csv -d $O/m$i -- $R/tools/microbench/gather2 > $O/m$i.log 2>&1
done
python3 - "$O" <<'PY'
import csv,glob,sys,collections,re
O=sys.argv[1]
for tag in ('b','m'):
    agg=collections.OrderedDict()
    for f in sorted(glob.glob(O+'/'+tag+'*/*/*_counter_collection.csv')):
        for r in csv.DictReader(open(f)):
            k=re.sub(r'^void ','',r['Kernel_Name'])[:60]
            if tag=='b' and 'fspt::' not in k: continue
            agg.setdefault((k,r['Counter_Name']),[]).append(float(r['Counter_Value']))
    with open(O+'/summary_'+tag+'.txt','w') as out:
        for (k,c),v in agg.items():
            out.write(f"{k:62s} {c:42s} launches={len(v):4d} sum={sum(v):.6g} max={max(v):.6g}\n")
PY
