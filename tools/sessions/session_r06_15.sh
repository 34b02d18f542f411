# usage (GPU box): bash tools/sessions/session_r06_15.sh
# the leaf records with the largest boxes in the trace kernel's LDS (WF_TRACE_LDS_LEAVES 0 / 2 / 4 / 8): what share of the leaf
# visits they serve (the counting variant), bit-equality, and the A/B
O=gpurun_out/r06; mkdir -p $O
FSPT_LIB=$PWD/ab_libs/h_hot4.so timeout 1200 python3 -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "intersect or render or fuzz or counters or chain or baseline or suspended or tail or two_level" > $O/gpu_hot_leaves_tests.log 2>&1; tail -2 $O/gpu_hot_leaves_tests.log
for n in h_head h_hot0 h_hot2 h_hot4 h_hot8; do echo -n "$n: "; FSPT_LIB=$PWD/ab_libs/$n.so timeout 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs --no-l1-microbench 2>/dev/null | python3 -c "
import sys,json
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); k=d['roofline']['kernels']['trace']
        print(d['value'], 'trace', k['ms_per_step'], 'lds nodes', k.get('lds_served_interior_steps'), 'lds leaves', k.get('lds_served_leaf_visits'), 'parity', d['parity_check']['equal'])
"; done > $O/hot_leaves_share.log 2>&1; cat $O/hot_leaves_share.log
bash tools/ab_run.sh "--steps 20 --warmup 5" h_head h_hot0 h_hot2 h_hot4 h_hot8 > $O/ab_hot_leaves_c2_20.log 2>&1; cat $O/ab_hot_leaves_c2_20.log
bash tools/ab_run.sh "--steps 20 --warmup 5 --config c3" h_head h_hot0 h_hot4 > $O/ab_hot_leaves_c3.log 2>&1; cat $O/ab_hot_leaves_c3.log
