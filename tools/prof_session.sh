#!/bin/bash
# Profile session (run on the GPU box through gpurun):
#   tools/prof_session.sh <tag> [bench args...]     -> gpurun_out/<tag>_{kt,fetch,write,sq*}/ + logs
# kernel trace + stats of the default bench command, then PMC passes (FETCH_SIZE and WRITE_SIZE in separate runs, as the
# guide prescribes; SQ / TA / TD sets) over one timed batch.  rocprofv3 gets the program itself after `--`.
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
T=$1; shift
O=$R/gpurun_out
cd /tmp
# the request-rate peak bench.py prices the trace kernel against: measured ONCE here, outside the profiler (a process
# rocprofv3 started has the GPU initialised by the preloaded tool library before main() runs and must not start children)
L1=$($R/tools/microbench/l1_peak 2>/dev/null | python3 -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'): print(json.loads(l)['gather_lane_requests_per_s'])" | tail -1)
L1ARG=${L1:+--l1-peak $L1}
[ -z "$L1" ] && L1ARG=--no-l1-microbench
echo "l1_peak: $L1" > $O/${T}_l1_peak.txt
set -- $L1ARG "$@"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_kt -- python3 $R/bench.py --no-cpu-baseline --no-extra-configs ${KT_ARGS:---steps 20 --warmup 5} "$@" > $O/${T}_kt.log 2>&1
find $O/${T}_kt -name "*kernel_trace.csv" -size +8M -delete
# the counter passes run the primary launch in the form this (lightly profiled) run settled on: under --pmc the kernels'
# timings are distorted and the library's form tuner would measure the profiler, not the kernels
PF=$(python3 -c "import sys,json
for l in open('$O/${T}_kt.log'):
    if l.startswith('{'): print(json.loads(l)['config'].get('primary_form',{}).get('form',0))" | tail -1)
[ -n "$PF" ] && [ "$PF" != "0" ] && set -- --primary-form $PF "$@"
echo "primary form for the counter passes: ${PF:-tuner}" >> $O/${T}_l1_peak.txt
# traffic passes: the timed configuration itself (TRAFFIC_ARGS, default: the driver's 20-step regions - one warm-up
# region, whose first batch has no live-path statistics yet, and four timed ones; the counters sum all five)
TA=${TRAFFIC_ARGS:---steps 20 --warmup 20 --reps 4}
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --output-format csv -d $O/${T}_$c -- python3 $R/bench.py --no-cpu-baseline --no-extra-configs --no-parity-check $TA "$@" > $O/${T}_$c.log 2>&1
done
# request sizes behind FETCH_SIZE (profiles/r03/fetch_calib.json: every read request is a 128-byte line)
timeout 600 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $O/${T}_rdreq -- python3 $R/bench.py --no-cpu-baseline --no-extra-configs --no-parity-check $TA "$@" > $O/${T}_rdreq.log 2>&1
i=0
for set in \
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU" \
 "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM SQ_WAVES" \
 "TA_TA_BUSY_sum TD_TD_BUSY_sum GRBM_GUI_ACTIVE" \
 "TCP_PERF_SEL_TOTAL_HIT_LRU_READ TCP_PERF_SEL_TOTAL_MISS_LRU_READ TCP_PERF_SEL_TOTAL_MISS_EVICT_READ" \
 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --output-format csv -d $O/${T}_sq$i -- python3 $R/bench.py ${SQ_ARGS:---steps 20 --warmup 0 --reps 1} --no-cpu-baseline --no-extra-configs --no-parity-check "$@" > $O/${T}_sq$i.log 2>&1
done
echo prof done
