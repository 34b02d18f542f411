#!/bin/bash
# Per-LAUNCH cache counters of the wavefront kernels (the launches of one 20-tick batch are its rounds, in dispatch order):
#   tools/pmc_rounds.sh <tag-under-gpurun_out> [bench args...]   (FSPT_LIB selects an A/B library)
# L1 (TCP) accesses / requests passed on to L2, L2 (TCC) hits / misses, VALU lane use.  One rocprofv3 --pmc pass per set,
# nothing but counters in a pass; the program itself after `--`.
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
T=$1; shift
O=$R/gpurun_out/$T
mkdir -p $O
cd /tmp
i=0
for set in \
 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
 "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum" \
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VALU" \
 "TA_TA_BUSY_sum TD_TD_BUSY_sum GRBM_GUI_ACTIVE TA_FLAT_READ_WAVEFRONTS_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --output-format csv -d $O/p$i -- python3 $R/bench.py --steps 20 --warmup 0 --reps 1 --no-cpu-baseline --no-extra-configs --no-parity-check --no-l1-microbench "$@" > $O/p$i.log 2>&1
done
python3 $R/tools/pmc_rounds.py $O > $O/rounds.txt 2>&1
cat $O/rounds.txt
