#!/bin/bash
# Focused PMC passes for the wavefront kernels' memory behaviour (separate --pmc passes, no trace domains).
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1; shift
mkdir -p $O
cd /tmp
i=0
for set in \
 "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
 "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
 "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
 "TCC_TAG_STALL_sum TCC_BUSY_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_LEVEL_sum" \
 "TA_TA_BUSY_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
 "SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM" \
 "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $O/p$i -- python3 $R/bench.py --steps 32 --warmup 32 --no-cpu-baseline "$@" > $O/p$i.log 2>&1
  grep -il "error\|invalid\|not found" $O/p$i.log | head -1
done
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 $R/bench.py --steps 32 --warmup 32 --no-cpu-baseline "$@" > $O/kt.log 2>&1
python3 - "$O" <<'PY'
import csv,glob,sys,collections,re
O=sys.argv[1]
agg=collections.OrderedDict()
for f in sorted(glob.glob(O+'/p*/*/*_counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        m=re.search(r'(k_wf_\w+)(<[^>]*>)?', r['Kernel_Name'])
        if not m: continue
        k=m.group(0)
        agg.setdefault((k,r['Counter_Name']),[]).append(float(r['Counter_Value']))
with open(O+'/summary.txt','w') as out:
    for (k,c),v in agg.items():
        out.write(f"{k:32s} {c:40s} launches={len(v):4d} sum={sum(v):.6g}\n")
f=glob.glob(O+'/kt/*/*_kernel_trace.csv')
if f:
    rows=sorted(csv.DictReader(open(f[0])), key=lambda r:int(r['Start_Timestamp']))
    idx=[i for i,r in enumerate(rows) if 'k_wf_logic<false, true' in r['Kernel_Name']]
    with open(O+'/rounds.txt','w') as out:
        for r in rows[idx[-1]:idx[-1]+21]:
            n=r['Kernel_Name'].split('(')[0].replace('void fspt::','')
            out.write(f"{n:34s} {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:9.1f} us\n")
    print(open(O+'/rounds.txt').read())
PY
