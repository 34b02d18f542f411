#!/bin/bash
# Collect PMC counters for the bench kernel in separate passes (rocprofv3 --pmc only; no trace domains).
# usage: tools/pmc_passes.sh <outdir-under-gpurun_out> [bench args...]
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1; shift
mkdir -p $O
cd /tmp
i=0
for set in \
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU" \
 "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM SQ_WAVES" \
 "SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_BRANCH SQ_INSTS_SMEM" \
 "TCP_PERF_SEL_TOTAL_HIT_LRU_READ TCP_PERF_SEL_TOTAL_MISS_LRU_READ TCP_PERF_SEL_TOTAL_MISS_EVICT_READ" \
 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
 "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $O/p$i -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extra-configs "$@" > $O/p$i.log 2>&1
done
python3 - "$O" <<'PY'
import csv,glob,sys,collections,re
O=sys.argv[1]
agg=collections.OrderedDict()
for f in sorted(glob.glob(O+'/p*/*/*_counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        m=re.search(r'(k_wf_\w+|k_trace|k_camera)(<[^>]*>)?', r['Kernel_Name'])
        if not m: continue
        k=m.group(0)
        if 'true>' in k and 'k_wf' in k and ', false' not in k and k.endswith('<true>'): continue  # counting variants
        agg.setdefault((k,r['Counter_Name']),[]).append(float(r['Counter_Value']))
with open(O+'/summary.txt','w') as out:
    for (k,c),v in agg.items():
        out.write(f"{k:28s} {c:36s} launches={len(v):4d} sum={sum(v):.6g} mean={sum(v)/len(v):.6g}\n")
print(open(O+'/summary.txt').read())
PY
