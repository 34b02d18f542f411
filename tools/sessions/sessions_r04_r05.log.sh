# GPU-box sessions of rounds 4 and 5, concatenated (one log instead of 66 scripts).
# Each block below was one `gpurun` call: /usr/local/graft/bin/gpurun --timeout N -- 'bash tools/sessions/<name>'.
# Kept as a record of exactly what was run; what each measured and what came of it: profiles/archive (README of the round).
# Blocks whose first line says so belong to experiments that were NOT adopted (their sources are not in the tree).

######## session_r04_1.sh
set -x
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04/s1_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r04/s1_pytest.log
tail -5 gpurun_out/r04/s1_pytest.log
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r04/s1_bench20.log 2>&1; tail -c 600 gpurun_out/r04/s1_bench20.log
tools/ab_run.sh "--steps 20 --warmup 5" base t256w4 t256w5 t512w6 > gpurun_out/r04/s1_ab_primary_occ.log 2>&1; cat gpurun_out/r04/s1_ab_primary_occ.log
tools/ab_run.sh "--steps 20 --warmup 5 --config c3" base > gpurun_out/r04/s1_c3.log 2>&1; cat gpurun_out/r04/s1_c3.log
tools/ab_run.sh "--steps 1 --warmup 3" base > gpurun_out/r04/s1_tick1.log 2>&1; cat gpurun_out/r04/s1_tick1.log

######## session_r04_2.sh
set -x
O=gpurun_out/r04; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q -k "not full_size and not million and not 4k" > $O/s2_pytest.log 2>&1; tail -3 $O/s2_pytest.log
tools/ab_run.sh "--steps 20 --warmup 5" base p256t p512t p256b p128t p256tu > $O/ab_primary_ticket.log 2>&1; cat $O/ab_primary_ticket.log
tools/ab_run.sh "--steps 20 --warmup 5 --config c3" base p256t p256tu > $O/ab_primary_ticket_c3.log 2>&1; cat $O/ab_primary_ticket_c3.log
tools/ab_run.sh "--steps 1 --warmup 3" base p256t p256tu > $O/ab_primary_ticket_tick1.log 2>&1; cat $O/ab_primary_ticket_tick1.log
tools/ab_run.sh "--steps 128 --warmup 128" base p256t p256tu > $O/ab_primary_ticket_128.log 2>&1; cat $O/ab_primary_ticket_128.log

######## session_r04_3.sh
set -x
O=gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/s3_pytest.log 2>&1; tail -3 $O/s3_pytest.log
tools/ab_run.sh "--steps 1 --warmup 3" ts1e9 ts32 ts32p ts16 ts64 > $O/ab_tail_slice_tick1.log 2>&1; cat $O/ab_tail_slice_tick1.log
tools/ab_run.sh "--steps 20 --warmup 5 --config c3" ts1e9 ts32 ts32p ts16 ts64 > $O/ab_tail_slice_c3.log 2>&1; cat $O/ab_tail_slice_c3.log
tools/ab_run.sh "--steps 20 --warmup 5" ts1e9 ts32 ts32p ts16 ts64 > $O/ab_tail_slice.log 2>&1; cat $O/ab_tail_slice.log

######## session_r04_4.sh
O=gpurun_out/r04; mkdir -p $O
for n in tprof tprof1e9; do
echo "== $n tick1"; FSPT_LIB=$PWD/ab_libs/$n.so timeout 300 python3 bench.py --steps 1 --warmup 3 --reps 1 --no-cpu-baseline --no-l1-microbench --no-parity-check 2>&1 | grep -E "tailprof" | tail -12
echo "== $n c3 20"; FSPT_LIB=$PWD/ab_libs/$n.so timeout 300 python3 bench.py --steps 20 --warmup 5 --reps 1 --config c3 --no-cpu-baseline --no-l1-microbench --no-parity-check 2>&1 | grep -E "tailprof" | tail -10
done > $O/tail_prof.log 2>&1
cat $O/tail_prof.log

######## session_r04_5.sh
O=gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/s5_pytest.log 2>&1; tail -3 $O/s5_pytest.log
tools/ab_run.sh "--steps 1 --warmup 3" tp t1 t1s16 t1s64 > $O/ab_tail_lane_tick1.log 2>&1; cat $O/ab_tail_lane_tick1.log
tools/ab_run.sh "--steps 20 --warmup 5 --config c3" tp t1 t1s16 t1s64 > $O/ab_tail_lane_c3.log 2>&1; cat $O/ab_tail_lane_c3.log
tools/ab_run.sh "--steps 20 --warmup 5" tp t1 t1s16 t1s64 > $O/ab_tail_lane.log 2>&1; cat $O/ab_tail_lane.log
echo "== t1prof tick1"; FSPT_LIB=$PWD/ab_libs/t1prof.so timeout 300 python3 bench.py --steps 1 --warmup 3 --reps 1 --no-cpu-baseline --no-l1-microbench --no-parity-check 2>&1 | grep -E "^tailprof" | tail -8

######## session_r04_6.sh
O=gpurun_out/r04; mkdir -p $O
tools/ab_run.sh "--steps 20 --warmup 5 --no-parity-check" e0 e1 e2 > $O/exp_noenv.log 2>&1; cat $O/exp_noenv.log

######## session_r04_7.sh
O=gpurun_out/r04; mkdir -p $O
tools/ab_run.sh "--steps 20 --warmup 5" cur im8 im24 im32 lu4 lu16 top15 top63 > $O/ab_tunables_r04.log 2>&1; cat $O/ab_tunables_r04.log

######## session_r04_8.sh
O=gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/s8_pytest.log 2>&1; tail -3 $O/s8_pytest.log
tools/ab_run.sh "--steps 20 --warmup 5" r1 r2 r2s4 r2s16 r3 > $O/ab_primary_refill.log 2>&1; cat $O/ab_primary_refill.log
tools/ab_run.sh "--steps 20 --warmup 5 --config c3" r1 r2 r2s16 r3 > $O/ab_primary_refill_c3.log 2>&1; cat $O/ab_primary_refill_c3.log
tools/ab_run.sh "--steps 1 --warmup 3" r1 r2 > $O/ab_primary_refill_tick1.log 2>&1; cat $O/ab_primary_refill_tick1.log
tools/ab_run.sh "--steps 128 --warmup 128" r1 r2 > $O/ab_primary_refill_128.log 2>&1; cat $O/ab_primary_refill_128.log

######## session_r04_9.sh
O=gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "not full_size and not million and not 4k" > $O/s9_pytest.log 2>&1; tail -2 $O/s9_pytest.log
tools/ab_run.sh "--steps 20 --warmup 5" r1 r2b r2bs16 > $O/ab_primary_refill2.log 2>&1; cat $O/ab_primary_refill2.log
tools/ab_run.sh "--steps 20 --warmup 5 --config c3" r1 r2b r2bs16 > $O/ab_primary_refill2_c3.log 2>&1; cat $O/ab_primary_refill2_c3.log
tools/ab_run.sh "--steps 1 --warmup 3" r1 r2b > $O/ab_primary_refill2_tick1.log 2>&1; cat $O/ab_primary_refill2_tick1.log
tools/ab_run.sh "--steps 128 --warmup 128" r1 r2b > $O/ab_primary_refill2_128.log 2>&1; cat $O/ab_primary_refill2_128.log

######## session_r04_10.sh
O=gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/s10_pytest.log 2>&1; tail -12 $O/s10_pytest.log
for a in "--steps 20 --warmup 5" "--steps 20 --warmup 5 --config c3" "--steps 1 --warmup 3" "--steps 128 --warmup 128" "--steps 20 --warmup 5 --config c3 --primary-form 1" "--steps 20 --warmup 5 --config c3 --primary-form 2"; do
  echo "== bench.py $a"
  timeout 900 python3 bench.py $a --no-cpu-baseline --no-l1-microbench 2>/dev/null | python3 -c "
import sys,json
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); k=d['roofline'].get('kernels',{})
        print(d['value'], 'Msamples/s', d['ms_per_step'], 'ms/step', d['rep_ms_per_step'], {c:v['ms_per_step'] for c,v in k.items()}, d['config'].get('primary_form'), 'parity', (d.get('parity_check') or {}).get('equal'))
"
done > $O/s10_tuner.log 2>&1
cat $O/s10_tuner.log

######## session_r04_11.sh
O=gpurun_out/r04; mkdir -p $O
for i in 1 2 3; do
for a in "--steps 20 --warmup 5 --config c3" "--steps 20 --warmup 5"; do
  echo "== bench.py $a"
  timeout 900 python3 bench.py $a --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); k=d['roofline'].get('kernels',{})
        print(d['value'], d['ms_per_step'], d['rep_ms_per_step'], {c:v['ms_per_step'] for c,v in k.items()}, d['config'].get('primary_form'))
"
done; done > $O/s11_tuner_repeat.log 2>&1
cat $O/s11_tuner_repeat.log

######## session_r04_12.sh
# (experiment of round 4 that was NOT adopted: the sources it ran - primary form 3 / the tail kernel with an LDS copy of the top of the tree - are not in the tree; results: profiles/r04/README.md)
# forms of the primary launch: 1 fused, 2 fused + refill, 3 split (k_wf_camtrace + shading)
O=gpurun_out/r04; mkdir -p $O
{
timeout 600 python3 -m pytest tests/test_parity_gpu.py -q -x -k "primary_launch_forms or bench_configuration or two_call" 2>&1 | tail -3
for rep in 1 2; do
for cfg in "--steps 20 --warmup 5" "--steps 20 --warmup 5 --config c3" "--steps 1 --warmup 3 --batch 1" "--steps 128 --warmup 128"; do
for f in 1 2 3; do
  echo -n "== $cfg form $f: "
  timeout 900 python3 bench.py $cfg --primary-form $f --no-cpu-baseline --no-l1-microbench 2>/dev/null | python3 -c "
import sys,json
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); k=d['roofline'].get('kernels',{})
        print(d['value'], d['ms_per_step'], {c:v['ms_per_step'] for c,v in k.items()}, 'parity', (d.get('parity_check') or {}).get('equal'))
"
done; done; done
} > $O/s12_split.log 2>&1
cat $O/s12_split.log

######## session_r04_13.sh
# (experiment of round 4 that was NOT adopted: the sources it ran - primary form 3 / the tail kernel with an LDS copy of the top of the tree - are not in the tree; results: profiles/r04/README.md)
O=$GRAFT_REPO_ROOT/gpurun_out/r04; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for f in 3 1; do
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_f$f -- python3 $R/bench.py --steps 20 --warmup 5 --primary-form $f --no-cpu-baseline --no-l1-microbench --no-parity-check > /tmp/prof_f$f.log 2>&1
find /tmp/prof_f$f -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/s13_form${f}_kernel_stats.csv
tail -3 /tmp/prof_f$f.log | cut -c1-300
done
head -8 $O/s13_form3_kernel_stats.csv; head -6 $O/s13_form1_kernel_stats.csv

######## session_r04_14.sh
# (experiment of round 4 that was NOT adopted: the sources it ran - primary form 3 / the tail kernel with an LDS copy of the top of the tree - are not in the tree; results: profiles/r04/README.md)
O=$GRAFT_REPO_ROOT/gpurun_out/r04; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
i=0
for set in \
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU" \
 "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM" \
 "TA_TA_BUSY_sum TD_TD_BUSY_sum GRBM_GUI_ACTIVE" \
 "SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_MISC SQ_WAVES SQ_INSTS_SMEM"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --output-format csv -d /tmp/split_sq$i -- python3 $R/bench.py --steps 20 --warmup 0 --reps 1 --primary-form 3 --no-cpu-baseline --no-parity-check --no-l1-microbench > /tmp/split_sq$i.log 2>&1
  tail -2 /tmp/split_sq$i.log | cut -c1-200
done
python3 $R/tools/pmc_quick.py /tmp/split_sq1 /tmp/split_sq2 /tmp/split_sq3 /tmp/split_sq4 > $O/s14_split_pmc.txt 2>&1
cat $O/s14_split_pmc.txt

######## session_r04_15.sh
# (experiment of round 4 that was NOT adopted: the sources it ran - primary form 3 / the tail kernel with an LDS copy of the top of the tree - are not in the tree; results: profiles/r04/README.md)
O=gpurun_out/r04; mkdir -p $O
{
for cfg in "--steps 1 --warmup 3 --batch 1" "--steps 20 --warmup 5 --config c3" "--steps 20 --warmup 5"; do
  echo "#### $cfg"
  bash tools/ab_run.sh "$cfg" tail0 tail256 bfs1k tail1k
done
} > $O/s15_tail_lds_top.log 2>&1
cat $O/s15_tail_lds_top.log

######## session_r04_16.sh
# the round after which the tail kernel takes over: forced values against the adaptive rule, 1 M-triangle scene and C2
O=gpurun_out/r04; mkdir -p $O
{
for rep in 1 2; do
for cfg in "--steps 20 --warmup 5 --config c3" "--steps 20 --warmup 5"; do
for tl in -1 3 4 5 6 7 0; do
  echo -n "== $cfg tail $tl: "
  timeout 900 python3 bench.py $cfg --tail $tl --no-cpu-baseline --no-l1-microbench 2>/dev/null | python3 -c "
import sys,json
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); k=d['roofline'].get('kernels',{})
        print(d['value'], d['ms_per_step'], {c:v['ms_per_step'] for c,v in k.items()}, {c:v['launches'] for c,v in k.items()}, 'parity', (d.get('parity_check') or {}).get('equal'))
"
done; done; done
} > $O/s16_tail_round.log 2>&1
cat $O/s16_tail_round.log

######## session_r04_17.sh
# (A/B of round 4, not adopted: pre-decoded flat-colour material sets; result: profiles/r04/README.md)
# A/B: material texture sets of flat colours pre-decoded in the table (cdec) against decoding per shading event (base)
O=gpurun_out/r04; mkdir -p $O
{
for cfg in "--steps 20 --warmup 5" "--steps 20 --warmup 5 --config c3" "--steps 128 --warmup 128"; do
  echo "#### $cfg"
  bash tools/ab_run.sh "$cfg" base cdec
done
} > $O/s17_const_sets.log 2>&1
cat $O/s17_const_sets.log

######## session_r04_18.sh
# the tuner with split batches: the forms test, then the choice and the regions of C3 / C2 three times over
O=gpurun_out/r04; mkdir -p $O
{
timeout 900 python3 -m pytest tests/test_parity_gpu.py -q -x -k "primary_launch_forms or two_call or deferred or memory_limit or fuzz" 2>&1 | tail -3
for i in 1 2 3; do
for a in "--steps 20 --warmup 5 --config c3" "--steps 20 --warmup 5" "--steps 128 --warmup 128" "--steps 1 --warmup 3"; do
  echo "== bench.py $a"
  timeout 900 python3 bench.py $a --no-cpu-baseline --no-l1-microbench 2>/dev/null | python3 -c "
import sys,json
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); k=d['roofline'].get('kernels',{})
        print(d['value'], d['ms_per_step'], d['rep_ms_per_step'], {c:v['ms_per_step'] for c,v in k.items()}, d['config'].get('primary_form'), (d.get('parity_check') or {}).get('equal'))
"
done; done
} > $O/s18_tuner_split.log 2>&1
cat $O/s18_tuner_split.log

######## session_r04_19.sh
# both profile sessions (C2 + tests + configs, C3) in one call
bash tools/sessions/session_r04_prof.sh final
bash tools/sessions/session_r04_c3.sh

######## session_r04_20.sh
O=gpurun_out/r04; mkdir -p $O
timeout 1200 python3 -m pytest tests/test_parity_gpu.py -q -x -k "two_ranks_share" > $O/s20_share_gpu.log 2>&1; tail -15 $O/s20_share_gpu.log
timeout 600 python3 bench.py --gpus 2 --share-gpu --steps 20 --warmup 5 --no-l1-microbench > $O/s20_share_gpu_bench.log 2>&1; tail -c 1500 $O/s20_share_gpu_bench.log

######## session_r04_21.sh
# flakiness check: the whole GPU suite five times over on one box
O=gpurun_out/r04; mkdir -p $O
for i in 1 2 3 4 5; do timeout 900 python3 -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -2; done > $O/s21_gpu_suite_x5.log 2>&1
cat $O/s21_gpu_suite_x5.log

######## session_r04_22.sh
# (A/B of round 4, not adopted: the logic kernel with its finishing paths processed densely; result: profiles/r04/README.md)
# A/B: k_wf_logic with the finishing paths listed and processed densely too (densefin) against every thread finishing its own (base)
O=gpurun_out/r04; mkdir -p $O
{
FSPT_LIB=$PWD/ab_libs/densefin.so timeout 900 python3 -m pytest tests/test_parity_gpu.py -q -x -k "fuzz or refractive or textured or counters or suspended or bench_configuration" 2>&1 | tail -2
for cfg in "--steps 20 --warmup 5" "--steps 20 --warmup 5 --config c3" "--steps 128 --warmup 128" "--steps 20 --warmup 5 --textured"; do
  echo "#### $cfg"
  bash tools/ab_run.sh "$cfg" base densefin
done
} > $O/s22_logic_dense_fin.log 2>&1
cat $O/s22_logic_dense_fin.log

######## session_r04_23.sh
# A/B: the six throughput divisions of a shading event once behind the reflect / Lambert branches (cdiv) instead of in both (base)
O=gpurun_out/r04; mkdir -p $O
{
FSPT_LIB=$PWD/ab_libs/cdiv.so timeout 900 python3 -m pytest tests/test_parity_gpu.py -q -x -k "fuzz or refractive or textured or bench_configuration or megakernel or brdf" 2>&1 | tail -2
for cfg in "--steps 20 --warmup 5" "--steps 20 --warmup 5 --config c3" "--steps 128 --warmup 128" "--steps 20 --warmup 5 --textured"; do
  echo "#### $cfg"
  bash tools/ab_run.sh "$cfg" base cdiv
done
} > $O/s23_common_divisions.log 2>&1
cat $O/s23_common_divisions.log

######## session_r04_24.sh
# the three full bench lines once more, with profiles/hbm_traffic.json stamped for the sources they run
O=gpurun_out/r04; mkdir -p $O
timeout 600 python3 bench.py --steps 20 --warmup 5 > $O/final_bench_steps20.json.log 2>&1
timeout 600 python3 bench.py > $O/final_bench_default.json.log 2>&1
timeout 600 python3 bench.py --steps 20 --warmup 5 --config c3 > $O/c3_bench_steps20.json.log 2>&1
tail -c 200 $O/c3_bench_steps20.json.log

######## session_r04_25.sh
# soak on the final kernels: 1000 fuzz seeds (seeds 0-399 are profiles/r04/fuzz_soak_400_seeds.log's)
O=gpurun_out/r04; mkdir -p $O
FSPT_FUZZ_SEEDS=1000 timeout 3300 python3 -m pytest tests/test_parity_gpu.py -m gpu -q -k "fuzz_random_scenes" -p no:cacheprovider > $O/fuzz_soak_1000_seeds.log 2>&1; tail -3 $O/fuzz_soak_1000_seeds.log

######## session_r04_26.sh
# final: tests, both profile sessions, then the three full bench lines
bash tools/sessions/session_r04_prof.sh final
bash tools/sessions/session_r04_c3.sh

######## session_r04_27.sh
# (timing experiment of round 4, results discarded: shade_hit without its reflect branch; result: profiles/r04/README.md)
# TIMING EXPERIMENT (results discarded): shade_hit never takes its reflect branch (nospec) - the upper bound of what
# deferring the few specular lanes of a wave to a dense pass could save - against the real kernels (base)
O=gpurun_out/r04; mkdir -p $O
{
for cfg in "--steps 20 --warmup 5 --no-parity-check" "--steps 20 --warmup 5 --config c3 --no-parity-check"; do
  echo "#### $cfg"
  bash tools/ab_run.sh "$cfg" base nospec
done
} > $O/s27_exp_no_specular.log 2>&1
cat $O/s27_exp_no_specular.log

######## session_r04_28.sh
# (A/B of round 4, not adopted: reflecting lanes of a shading wave parked for a dense launch; result: profiles/r04/README.md)
# A/B: a wave's few reflecting lanes parked for k_wf_spec (defer) against shaded in place (nodefer)
O=gpurun_out/r04; mkdir -p $O
{
FSPT_LIB=$PWD/ab_libs/defer.so timeout 1500 python3 -m pytest tests/test_parity_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3
for cfg in "--steps 20 --warmup 5" "--steps 20 --warmup 5 --config c3" "--steps 128 --warmup 128"; do
  echo "#### $cfg"
  bash tools/ab_run.sh "$cfg" nodefer defer
done
} > $O/s28_defer_reflect.log 2>&1
cat $O/s28_defer_reflect.log

######## session_r04_29.sh
# what the driver runs at round end, on the final tree: smoke(), the GPU suite, the bench command
O=gpurun_out/r04; mkdir -p $O
{
timeout 600 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
timeout 900 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -2
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | cut -c1-400
} > $O/s29_driver_like.log 2>&1
cat $O/s29_driver_like.log

######## session_r04_30.sh
# the 4- and 8-rank programs on ONE GPU (--share-gpu: gloo through host memory): plumbing of the weak-scaled frames
# (3840x2160, 5432x3056), tile dealing over 4 / 8 ranks, the gather of 4 / 8 pieces, parity of the assembled frame
O=gpurun_out/r04; mkdir -p $O
{
for n in 4 8; do
  echo "== bench.py --gpus $n --share-gpu --steps 8 --warmup 2 --reps 2"
  timeout 1200 python3 bench.py --gpus $n --share-gpu --steps 8 --warmup 2 --reps 2 --no-l1-microbench --rendezvous-timeout 300 2>$O/s30_share_gpu_n$n.err | python3 -c "
import sys,json
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line)
        print(d['metric'], d['value'], d['n_gpus'], d['config']['world_size_seen'], d['config']['sharding'], 'exchange_ms', d.get('exchange_ms'), 'parity', d['parity_check'])
"
  grep -c "bench rank" $O/s30_share_gpu_n$n.err
done
} > $O/s30_share_gpu_4_8.log 2>&1
cat $O/s30_share_gpu_4_8.log

######## session_r04_31.sh
# BASELINE configs[3] (C4: 3840x2160, depth 8, 8 ranks, strong scaling) as an 8-rank program on ONE GPU, both exchanges
O=gpurun_out/r04; mkdir -p $O
{
for ex in gather reduce; do
  echo "== bench.py --gpus 8 --share-gpu --config c4 --exchange $ex --steps 8 --warmup 2 --reps 2"
  timeout 1200 python3 bench.py --gpus 8 --share-gpu --config c4 --exchange $ex --steps 8 --warmup 2 --reps 2 --no-l1-microbench --rendezvous-timeout 300 2>$O/s31_c4_$ex.err | python3 -c "
import sys,json
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line)
        print(d['metric'], d['value'], d['scaling'], d['n_gpus'], d['config']['world_size_seen'], d['config']['sharding'], d['config']['exchange'][:20], 'exchange_ms', d.get('exchange_ms'), 'parity', d['parity_check'])
"
  grep -c "bench rank" $O/s31_c4_$ex.err
done
} > $O/s31_share_gpu_c4.log 2>&1
cat $O/s31_share_gpu_c4.log

######## session_r04_32.sh
# the suspension budget of the trace launches (steps a starved wave walks on before it parks its rays) on the final
# kernels: the default against a few values, C2 and the 1 M-triangle scene at 20 steps
O=gpurun_out/r04; mkdir -p $O
{
for rep in 1 2; do
for cfg in "--steps 20 --warmup 5" "--steps 20 --warmup 5 --config c3"; do
for tb in -1 0 8 16 32 64 128; do
  echo -n "== $cfg budget $tb: "
  timeout 900 python3 bench.py $cfg --trace-budget $tb --no-cpu-baseline --no-l1-microbench 2>/dev/null | python3 -c "
import sys,json
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); k=d['roofline'].get('kernels',{})
        print(d['value'], d['ms_per_step'], {c:v['ms_per_step'] for c,v in k.items()}, 'parity', (d.get('parity_check') or {}).get('equal'))
"
done; done; done
} > $O/s32_trace_budget.log 2>&1
cat $O/s32_trace_budget.log

######## session_r04_33.sh
# A/B: the last trace launch in front of the tail kernel suspends its long rays too, the tail kernel traces them again
# from the root (lastsusp), against letting them finish in the trace launch (base)
O=gpurun_out/r04; mkdir -p $O
{
FSPT_LIB=$PWD/ab_libs/lastsusp.so timeout 900 python3 -m pytest tests/test_parity_gpu.py -q -x -k "fuzz or refractive or suspended or bench_configuration or million" 2>&1 | tail -2
for cfg in "--steps 20 --warmup 5" "--steps 20 --warmup 5 --config c3" "--steps 1 --warmup 3"; do
  echo "#### $cfg"
  bash tools/ab_run.sh "$cfg" base lastsusp
done
} > $O/s33_last_trace_suspends.log 2>&1
cat $O/s33_last_trace_suspends.log

######## session_r04_34.sh
# soak after the last change (the last trace launch suspends too): 200 fuzz seeds
O=gpurun_out/r04; mkdir -p $O
FSPT_FUZZ_SEEDS=200 timeout 1200 python3 -m pytest tests/test_parity_gpu.py -m gpu -q -k "fuzz_random_scenes" -p no:cacheprovider > $O/fuzz_soak_200_seeds_final.log 2>&1; tail -2 $O/fuzz_soak_200_seeds_final.log

######## session_r04_35.sh
# the tail hand-over round again, now that the last trace launch suspends: single tick and the 1 M-triangle scene
O=gpurun_out/r04; mkdir -p $O
{
for cfg in "--steps 1 --warmup 3" "--steps 20 --warmup 5 --config c3"; do
for tl in -1 1 2 3 4 5 6; do
  echo -n "== $cfg tail $tl: "
  timeout 900 python3 bench.py $cfg --tail $tl --no-cpu-baseline --no-l1-microbench 2>/dev/null | python3 -c "
import sys,json
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); k=d['roofline'].get('kernels',{})
        print(d['value'], d['ms_per_step'], {c:v['ms_per_step'] for c,v in k.items()}, {c:v['launches'] for c,v in k.items()}, 'parity', (d.get('parity_check') or {}).get('equal'))
"
done; done
} > $O/s35_tail_round_after.log 2>&1
cat $O/s35_tail_round_after.log

######## session_r04_c3.sh
O=gpurun_out/r04; mkdir -p $O
bash tools/prof_r04.sh r04/c3 --config c3 > $O/c3_prof.log 2>&1; tail -1 $O/c3_prof.log
timeout 600 python3 bench.py --steps 20 --warmup 5 --config c3 > $O/c3_bench_steps20.json.log 2>&1; tail -c 300 $O/c3_bench_steps20.json.log

######## session_r04_prof.sh
# usage (GPU box): bash tools/sessions/session_r04_prof.sh <tag>
T=${1:-final}
O=gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/${T}_pytest.log 2>&1; tail -3 $O/${T}_pytest.log
bash tools/prof_r04.sh r04/$T > $O/${T}_prof.log 2>&1; tail -2 $O/${T}_prof.log
timeout 600 python3 bench.py --steps 20 --warmup 5 > $O/${T}_bench_steps20.json.log 2>&1
timeout 600 python3 bench.py > $O/${T}_bench_default.json.log 2>&1
{ for a in "--steps 20 --warmup 5" "--steps 128 --warmup 128" "--steps 1 --warmup 3" "--steps 20 --warmup 5 --config c3" "--steps 128 --warmup 128 --config c3" "--steps 20 --warmup 5 --config c5" "--steps 20 --warmup 5 --textured" "--steps 20 --warmup 5 --width 3840 --height 2160" "--steps 20 --warmup 5 --pipeline stream" "--steps 4 --warmup 2 --pipeline megakernel"; do
  echo "== bench.py $a"
  timeout 900 python3 bench.py $a --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); k=d['roofline'].get('kernels',{})
        print(d['value'], 'Msamples/s', d['ms_per_step'], 'ms/step', {c:v['ms_per_step'] for c,v in k.items()}, 'parity', (d.get('parity_check') or {}).get('equal'))
"
done; } > $O/${T}_configs_one_box.log 2>&1
cat $O/${T}_configs_one_box.log

######## session_r04_soak.sh
O=gpurun_out/r04; mkdir -p $O
FSPT_FUZZ_SEEDS=400 timeout 2400 python -m pytest tests/test_parity_gpu.py -m gpu -q -k "fuzz_random_scenes" -p no:cacheprovider > $O/fuzz_soak_400_seeds.log 2>&1; tail -3 $O/fuzz_soak_400_seeds.log

######## session_r05_1.sh
O=gpurun_out/r05; mkdir -p $O
timeout 1500 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "two_level or deep_chain or fuzz or not_unions or intersect" -p no:cacheprovider > $O/tests_two_level.log 2>&1; tail -5 $O/tests_two_level.log
V=("--node-form 0,0,0" "--node-form 0,0,1" "--node-form 0,1,1" "--node-form 1,0,1" "--node-form 0,-1,1,300000" "--node-form 0,-1,1,1500000")
tools/ab_args.sh "--steps 20 --warmup 5" "${V[@]}" > $O/ab_two_level_c2_20.log 2>&1; cat $O/ab_two_level_c2_20.log
tools/ab_args.sh "--steps 1 --warmup 3 --reps 9" "${V[@]}" > $O/ab_two_level_c2_tick1.log 2>&1; cat $O/ab_two_level_c2_tick1.log
tools/ab_args.sh "--config c3 --steps 20 --warmup 5" "--node-form 0,0,0" "--node-form 0,0,1" "--node-form 0,1,1" "--node-form 1,1,1" "--node-form 1,0,1" > $O/ab_two_level_c3_20.log 2>&1; cat $O/ab_two_level_c3_20.log

######## session_r05_2.sh
O=gpurun_out/r05; mkdir -p $O
tools/ab_run.sh "--steps 20 --warmup 5" r5base r5carry r5clears r5both > $O/ab_fixed_costs_c2_20.log 2>&1; cat $O/ab_fixed_costs_c2_20.log
tools/ab_run.sh "--steps 1 --warmup 3 --reps 9" r5base r5both > $O/ab_fixed_costs_tick1.log 2>&1; cat $O/ab_fixed_costs_tick1.log
tools/ab_run.sh "--config c3 --steps 20 --warmup 5" r5base r5both > $O/ab_fixed_costs_c3.log 2>&1; cat $O/ab_fixed_costs_c3.log
timeout 2400 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/gpu_suite_1.log 2>&1; tail -5 $O/gpu_suite_1.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_default_extra.json.log 2> $O/bench_default_extra.err; tail -c 3000 $O/bench_default_extra.json.log; tail -3 $O/bench_default_extra.err

######## session_r05_3.sh
O=gpurun_out/r05; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "rccl or pack_and_unpack or close_executes or tile_gather or share_the_gpu or multi_device or two_level" > $O/gpu_new_tests.log 2>&1; tail -5 $O/gpu_new_tests.log
tools/ab_run.sh "--steps 20 --warmup 5" r5c0 r5c32 r5c64 r5sinf32 > $O/ab_carry_sin_c2_20.log 2>&1; cat $O/ab_carry_sin_c2_20.log
tools/ab_run.sh "--config c3 --steps 20 --warmup 5" r5c0 r5c32 r5sinf32 > $O/ab_carry_sin_c3.log 2>&1; cat $O/ab_carry_sin_c3.log

######## session_r05_4.sh
O=gpurun_out/r05; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "js_bench or halved or stream_pool or memory_limit or path_state" > $O/gpu_new_tests_2.log 2>&1; tail -5 $O/gpu_new_tests_2.log
cd /tmp
for c in c2 c3; do
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$O/kt_$c -- python3 $R/bench.py --config $c --steps 20 --warmup 5 --reps 3 --no-cpu-baseline --no-extra-configs --no-l1-microbench --no-parity-check > $R/$O/kt_$c.log 2>&1
  python3 $R/tools/launch_list.py $R/$O/kt_$c > $R/$O/launch_list_$c.txt 2>&1; tail -40 $R/$O/launch_list_$c.txt
  find $R/$O/kt_$c -name "*kernel_trace.csv" -size +4M -delete
done
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$O/kt_tick1 -- python3 $R/bench.py --steps 1 --warmup 3 --reps 3 --no-cpu-baseline --no-extra-configs --no-l1-microbench --no-parity-check > $R/$O/kt_tick1.log 2>&1
python3 $R/tools/launch_list.py $R/$O/kt_tick1 > $R/$O/launch_list_tick1.txt 2>&1; tail -30 $R/$O/launch_list_tick1.txt

######## session_r05_5.sh
O=gpurun_out/r05; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "two_level or tail or halved or stream_pool or js_bench or fuzz or deferred or two_call" > $O/gpu_new_tests_3.log 2>&1; tail -4 $O/gpu_new_tests_3.log
tools/ab_run.sh "--steps 20 --warmup 5" r5ownstream r5tstream > $O/ab_target_stream_c2_20.log 2>&1; cat $O/ab_target_stream_c2_20.log
tools/ab_run.sh "--steps 1 --warmup 3 --reps 9" r5ownstream r5tstream > $O/ab_target_stream_tick1.log 2>&1; cat $O/ab_target_stream_tick1.log
V=("--node-form 0,0,0" "--node-form 0,0,2" "--node-form 0,0,1")
tools/ab_args.sh "--steps 20 --warmup 5" "${V[@]}" > $O/ab_tail_adaptive_c2_20.log 2>&1; cat $O/ab_tail_adaptive_c2_20.log
tools/ab_args.sh "--steps 1 --warmup 3 --reps 9" "${V[@]}" > $O/ab_tail_adaptive_tick1.log 2>&1; cat $O/ab_tail_adaptive_tick1.log
tools/ab_args.sh "--config c3 --steps 20 --warmup 5" "${V[@]}" > $O/ab_tail_adaptive_c3.log 2>&1; cat $O/ab_tail_adaptive_c3.log
echo "== stage events off"; for i in 1 2; do FSPT_STAGE_EVENTS=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs --no-l1-microbench 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('events off', d['value'], d['ms_per_step'])"; python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs --no-l1-microbench 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('events on ', d['value'], d['ms_per_step'])"; done > $O/ab_stage_events.log 2>&1; cat $O/ab_stage_events.log

######## session_r05_6.sh
O=gpurun_out/r05; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/gpu_suite_2.log 2>&1; tail -4 $O/gpu_suite_2.log
for i in 1 2; do timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_driver_form_$i.json.log 2>/dev/null; python3 -c "
import json
for l in open('$O/bench_driver_form_$i.json.log'):
    if l.startswith('{'):
        d=json.loads(l); print('driver form', d['value'], d['ms_per_step'], d['rep_ms_per_step'], d['stage_events']['region_ms_per_step'], {c:v['ms_per_step'] for c,v in d['roofline']['kernels'].items()}, {k:(v['value'], v['parity_check']['equal']) for k,v in d.get('extra_configs',{}).items()}, d['parity_check']['equal'])"; done
timeout 900 python bench.py --no-extra-configs > $O/bench_default_128.json.log 2>/dev/null; python3 -c "
import json
for l in open('$O/bench_default_128.json.log'):
    if l.startswith('{'):
        d=json.loads(l); print('128 steps', d['value'], d['ms_per_step'], {c:v['ms_per_step'] for c,v in d['roofline']['kernels'].items()})"
tools/ab_args.sh "--steps 1 --warmup 3 --reps 9" "--stage-events all" "--stage-events last" > $O/bench_tick1.log 2>&1; cat $O/bench_tick1.log
tools/ab_args.sh "--steps 20 --warmup 5" "--pipeline stream" "--textured" "--width 3840 --height 2160" > $O/bench_other_configs.log 2>&1; cat $O/bench_other_configs.log
python tools/write_bench_scene.py /tmp/benchscene > /dev/null && node fspt_amd/js/bench.js --scene /tmp/benchscene/scene/bench.json --focal-depth 2 --aperture 0.02 --steps 20 --warmup 5 > $O/bench_node_host.json.log 2>&1; cat $O/bench_node_host.json.log | cut -c1-400
FSPT_FUZZ_SEEDS=400 timeout 2400 python -m pytest tests/test_parity_gpu.py -m gpu -q -k "fuzz_random_scenes" -p no:cacheprovider > $O/fuzz_soak_400_seeds.log 2>&1; tail -3 $O/fuzz_soak_400_seeds.log

######## session_r05_7.sh
O=gpurun_out/r05; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "bench or share_the_gpu or eight_ranks or rccl or js_" > $O/gpu_bench_tests.log 2>&1; grep -E "passed|failed" $O/gpu_bench_tests.log | tail -2
bash tools/sessions/session_r05_prof.sh

######## session_r05_8.sh
O=gpurun_out/r05; mkdir -p $O
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 2400 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/gpu_suite_3.log 2>&1; grep -E "passed|failed" $O/gpu_suite_3.log | tail -2
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_bench.json.log 2> $O/driver_bench.err; python3 -c "
import json
for l in open('$O/driver_bench.json.log'):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print(d['value'], d['ms_per_step'], d['rep_ms_per_step'], 'frac', r['frac'], 'hbm_counter', r['hbm_counter'] and r['hbm_counter']['frac'], 'primary', r['kernels']['primary'].get('frac'), 'blend c3', d['extra_configs']['c3']['roofline'].get('blended_with_l2_hit_rate'), {k:(v['value'],v['parity_check']['equal']) for k,v in d['extra_configs'].items()}, d['parity_check']['equal'], d['cpu_baseline']['value'])"

######## session_r05_9.sh
O=gpurun_out/r05; mkdir -p $O
tools/ab_run.sh "--steps 20 --warmup 5" r5b7 r5b6t16 r5b6t31 r5b6t63 > $O/ab_trace_lds_top_c2_20.log 2>&1; cat $O/ab_trace_lds_top_c2_20.log
tools/ab_run.sh "--steps 128 --warmup 128" r5b7 r5b6t31 r5b6t63 > $O/ab_trace_lds_top_c2_128.log 2>&1; cat $O/ab_trace_lds_top_c2_128.log
tools/ab_run.sh "--config c3 --steps 20 --warmup 5" r5b7 r5b6t31 r5b6t63 > $O/ab_trace_lds_top_c3.log 2>&1; cat $O/ab_trace_lds_top_c3.log

######## session_r05_10.sh
O=gpurun_out/r05; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=20 > $O/gpu_suite_durations.log 2>&1; grep -E "passed|failed|s call|s setup" $O/gpu_suite_durations.log | tail -24
for i in 1 2 3; do timeout 2400 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/gpu_suite_x3_$i.log 2>&1; grep -E "passed|failed" $O/gpu_suite_x3_$i.log | tail -1; done
FSPT_FUZZ_SEEDS=1000 timeout 3000 python -m pytest tests/test_parity_gpu.py -m gpu -q -k "fuzz_random_scenes" -p no:cacheprovider > $O/fuzz_soak_1000_seeds.log 2>&1; grep -E "passed|failed" $O/fuzz_soak_1000_seeds.log | tail -1

######## session_r05_11.sh
O=gpurun_out/r05; mkdir -p $O
tools/ab_run.sh "--steps 20 --warmup 5 --tail 8" r5p32 r5p8 r5p2 r5p1 > $O/ab_tail_pairs_c2_20_tail8.log 2>&1; cat $O/ab_tail_pairs_c2_20_tail8.log
tools/ab_run.sh "--steps 20 --warmup 5 --tail 6" r5p32 r5p8 r5p2 > $O/ab_tail_pairs_c2_20_tail6.log 2>&1; cat $O/ab_tail_pairs_c2_20_tail6.log
tools/ab_run.sh "--steps 1 --warmup 3 --reps 9 --tail 6" r5p32 r5p8 r5p2 r5p1 > $O/ab_tail_pairs_tick1_tail6.log 2>&1; cat $O/ab_tail_pairs_tick1_tail6.log
tools/ab_run.sh "--config c3 --steps 20 --warmup 5 --tail 8" r5p32 r5p8 r5p2 r5p1 > $O/ab_tail_pairs_c3_tail8.log 2>&1; cat $O/ab_tail_pairs_c3_tail8.log

######## session_r05_12.sh
O=gpurun_out/r05; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "tail or fuzz or deep_chain or stream or refractive" > $O/gpu_tail_tests.log 2>&1; grep -E "passed|failed" $O/gpu_tail_tests.log | tail -1
tools/ab_args.sh "--steps 20 --warmup 5" "--tail -1" "--tail 4" "--tail 5" "--tail 6" "--tail 7" "--tail 8" > $O/scan_tail_round_auto_pairs_c2_20.log 2>&1; cat $O/scan_tail_round_auto_pairs_c2_20.log
tools/ab_args.sh "--steps 1 --warmup 3 --reps 9" "--tail -1" "--tail 2" "--tail 3" "--tail 4" "--tail 5" "--tail 6" > $O/scan_tail_round_auto_pairs_tick1.log 2>&1; cat $O/scan_tail_round_auto_pairs_tick1.log
tools/ab_args.sh "--config c3 --steps 20 --warmup 5" "--tail -1" "--tail 5" "--tail 6" "--tail 7" "--tail 8" > $O/scan_tail_round_auto_pairs_c3.log 2>&1; cat $O/scan_tail_round_auto_pairs_c3.log
tools/ab_args.sh "--steps 4 --warmup 4" "--tail -1" "--tail 3" "--tail 4" "--tail 5" "--tail 6" > $O/scan_tail_round_auto_pairs_c2_4.log 2>&1; cat $O/scan_tail_round_auto_pairs_c2_4.log

######## session_r05_13.sh
O=gpurun_out/r05; mkdir -p $O
tools/ab_run.sh "--width 480 --height 270 --steps 1 --warmup 3 --reps 15" r5pairs32 r5pairsauto > $O/ab_tail_pairs_auto_480x270_tick1.log 2>&1; cat $O/ab_tail_pairs_auto_480x270_tick1.log
tools/ab_run.sh "--width 960 --height 540 --steps 1 --warmup 3 --reps 15" r5pairs32 r5pairsauto > $O/ab_tail_pairs_auto_960x540_tick1.log 2>&1; cat $O/ab_tail_pairs_auto_960x540_tick1.log
tools/ab_run.sh "--width 480 --height 270 --steps 20 --warmup 5" r5pairs32 r5pairsauto > $O/ab_tail_pairs_auto_480x270_20.log 2>&1; cat $O/ab_tail_pairs_auto_480x270_20.log
tools/ab_run.sh "--steps 20 --warmup 5" r5pairs32 r5pairsauto > $O/ab_tail_pairs_auto_c2_20.log 2>&1; cat $O/ab_tail_pairs_auto_c2_20.log
rm -rf $O/final_* $O/c3_* $O/c5_*
timeout 1800 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/gpu_suite_final.log 2>&1; grep -E "passed|failed" $O/gpu_suite_final.log | tail -1
bash tools/sessions/session_r05_prof.sh

######## session_r05_14.sh
O=gpurun_out/r05; mkdir -p $O
FSPT_LIB=$PWD/ab_libs/r5xcd.so timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -p no:cacheprovider -k "render_fused or suspended or fuzz or two_call or refractive or full_size or million" > $O/gpu_xcd_tests.log 2>&1; grep -E "passed|failed" $O/gpu_xcd_tests.log | tail -1
tools/ab_run.sh "--steps 20 --warmup 5" r5base r5xcd > $O/ab_trace_xcd_c2_20.log 2>&1; cat $O/ab_trace_xcd_c2_20.log
tools/ab_run.sh "--steps 128 --warmup 128" r5base r5xcd > $O/ab_trace_xcd_c2_128.log 2>&1; cat $O/ab_trace_xcd_c2_128.log
tools/ab_run.sh "--config c3 --steps 20 --warmup 5" r5base r5xcd > $O/ab_trace_xcd_c3.log 2>&1; cat $O/ab_trace_xcd_c3.log
tools/ab_run.sh "--steps 20 --warmup 5 --width 3840 --height 2160" r5base r5xcd > $O/ab_trace_xcd_4k.log 2>&1; cat $O/ab_trace_xcd_4k.log

######## session_r05_15.sh
O=gpurun_out/r05; mkdir -p $O
FSPT_LIB=$PWD/ab_libs/r5xnackoff.so timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -p no:cacheprovider -k "render_fused or suspended or fuzz or two_call or refractive" > $O/gpu_xnackoff_tests.log 2>&1; grep -E "passed|failed|rror" $O/gpu_xnackoff_tests.log | tail -2
tools/ab_run.sh "--steps 20 --warmup 5" r5base r5xnackoff r5ilp r5memcl > $O/ab_flags_c2_20.log 2>&1; cat $O/ab_flags_c2_20.log
tools/ab_run.sh "--steps 128 --warmup 128" r5base r5xnackoff r5ilp r5memcl > $O/ab_flags_c2_128.log 2>&1; cat $O/ab_flags_c2_128.log
tools/ab_run.sh "--config c3 --steps 20 --warmup 5" r5base r5xnackoff r5ilp r5memcl > $O/ab_flags_c3.log 2>&1; cat $O/ab_flags_c3.log

######## session_r05_16.sh
O=gpurun_out/r05; mkdir -p $O
FSPT_LIB=$PWD/ab_libs/r5pf2.so timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -p no:cacheprovider -k "render_fused or suspended or fuzz or two_call or refractive" > $O/gpu_tail_prefetch_tests.log 2>&1; grep -E "passed|failed|rror" $O/gpu_tail_prefetch_tests.log | tail -2
tools/ab_run.sh "--steps 1 --warmup 3 --reps 9" r5base r5pf1 r5pf2 > $O/ab_tail_prefetch_tick1.log 2>&1; cat $O/ab_tail_prefetch_tick1.log
tools/ab_run.sh "--steps 20 --warmup 5" r5base r5pf1 r5pf2 > $O/ab_tail_prefetch_c2_20.log 2>&1; cat $O/ab_tail_prefetch_c2_20.log
tools/ab_run.sh "--config c3 --steps 20 --warmup 5" r5base r5pf1 r5pf2 > $O/ab_tail_prefetch_c3.log 2>&1; cat $O/ab_tail_prefetch_c3.log

######## session_r05_17.sh
O=gpurun_out/r05; mkdir -p $O
FSPT_LIB=$PWD/ab_libs/r5pin.so timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -p no:cacheprovider -k "render_fused or suspended or fuzz or two_call or refractive" > $O/gpu_pin_tests.log 2>&1; grep -E "passed|failed|rror" $O/gpu_pin_tests.log | tail -2
tools/ab_run.sh "--steps 1 --warmup 3 --reps 9" r5base r5pin > $O/ab_pin_pointers_tick1.log 2>&1; cat $O/ab_pin_pointers_tick1.log
tools/ab_run.sh "--steps 20 --warmup 5" r5base r5pin > $O/ab_pin_pointers_c2_20.log 2>&1; cat $O/ab_pin_pointers_c2_20.log
tools/ab_run.sh "--config c3 --steps 20 --warmup 5" r5base r5pin > $O/ab_pin_pointers_c3.log 2>&1; cat $O/ab_pin_pointers_c3.log
tools/ab_run.sh "--steps 128 --warmup 128" r5base r5pin > $O/ab_pin_pointers_c2_128.log 2>&1; cat $O/ab_pin_pointers_c2_128.log

######## session_r05_18.sh
O=gpurun_out/r05; mkdir -p $O
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -p no:cacheprovider -k "tail_stages or refractive" > $O/gpu_tail_stages_tests.log 2>&1; grep -E "passed|failed|rror" $O/gpu_tail_stages_tests.log | tail -3
V=("--tail-stages 0" "--tail-stages 32" "--tail-stages 32,4" "--tail-stages 32,8" "--tail-stages 16" "--tail-stages 8" "--tail-stages 16,2")
tools/ab_args.sh "--steps 1 --warmup 3 --reps 9" "${V[@]}" > $O/ab_tail_stages_tick1.log 2>&1; cat $O/ab_tail_stages_tick1.log
tools/ab_args.sh "--steps 20 --warmup 5" "${V[@]}" > $O/ab_tail_stages_c2_20.log 2>&1; cat $O/ab_tail_stages_c2_20.log
tools/ab_args.sh "--config c3 --steps 20 --warmup 5" "${V[@]}" > $O/ab_tail_stages_c3.log 2>&1; cat $O/ab_tail_stages_c3.log

######## session_r05_19.sh
O=gpurun_out/r05; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/gpu_suite_head.log 2>&1; tail -2 $O/gpu_suite_head.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/head_bench_steps20.json.log 2>$O/head_bench_steps20.err; python3 -c "
import json
for l in open('$O/head_bench_steps20.json.log'):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print(d['value'], r['frac'], 'traffic', r['traffic'], r.get('traffic_source'), {k:v['value'] for k,v in d['extra_configs'].items()})
"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1

######## session_r05_20.sh
O=gpurun_out/r05; mkdir -p $O
tools/ab_run.sh "--steps 20 --warmup 5" r5base r5envnt r5hitnt > $O/ab_nontemporal_c2_20.log 2>&1; cat $O/ab_nontemporal_c2_20.log
tools/ab_run.sh "--config c3 --steps 20 --warmup 5" r5base r5envnt r5hitnt > $O/ab_nontemporal_c3.log 2>&1; cat $O/ab_nontemporal_c3.log

######## session_r05_21.sh
O=gpurun_out/r05; mkdir -p $O
FSPT_LIB=$PWD/ab_libs/r5tickets.so timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -p no:cacheprovider -k "render_fused or primary_launch or fuzz or two_call or ragged or viewport or stream_scheduler" > $O/gpu_tickets_tests.log 2>&1; grep -E "passed|failed|rror" $O/gpu_tickets_tests.log | tail -2
tools/ab_run.sh "--steps 20 --warmup 5" r5static r5tickets > $O/ab_primary_tickets_c2_20.log 2>&1; cat $O/ab_primary_tickets_c2_20.log
tools/ab_run.sh "--config c3 --steps 20 --warmup 5" r5static r5tickets > $O/ab_primary_tickets_c3.log 2>&1; cat $O/ab_primary_tickets_c3.log
tools/ab_run.sh "--steps 128 --warmup 128" r5static r5tickets > $O/ab_primary_tickets_c2_128.log 2>&1; cat $O/ab_primary_tickets_c2_128.log
tools/ab_run.sh "--steps 1 --warmup 3 --reps 9" r5static r5tickets > $O/ab_primary_tickets_tick1.log 2>&1; cat $O/ab_primary_tickets_tick1.log

######## session_r05_22.sh
O=gpurun_out/r05; mkdir -p $O
FSPT_LIB=$PWD/ab_libs/r5tickets2.so timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -p no:cacheprovider -k "render_fused or primary_launch or fuzz or two_call or ragged or viewport or stream_scheduler or suspended or refractive or tail_kernel" > $O/gpu_tickets2_tests.log 2>&1; grep -E "passed|failed|rror" $O/gpu_tickets2_tests.log | tail -2
tools/ab_run.sh "--steps 20 --warmup 5" r5static r5tickets r5tickets2 > $O/ab_logic_tickets_c2_20.log 2>&1; cat $O/ab_logic_tickets_c2_20.log
tools/ab_run.sh "--config c3 --steps 20 --warmup 5" r5static r5tickets r5tickets2 > $O/ab_logic_tickets_c3.log 2>&1; cat $O/ab_logic_tickets_c3.log
tools/ab_run.sh "--steps 128 --warmup 128" r5tickets r5tickets2 > $O/ab_logic_tickets_c2_128.log 2>&1; cat $O/ab_logic_tickets_c2_128.log

######## session_r05_23.sh
O=gpurun_out/r05; mkdir -p $O
FSPT_LIB=$PWD/ab_libs/r5tickets3.so timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -p no:cacheprovider -k "render_fused or primary_launch or fuzz or two_call or ragged or viewport or stream_scheduler or suspended or refractive or tail_kernel" > $O/gpu_tickets3_tests.log 2>&1; grep -E "passed|failed|rror" $O/gpu_tickets3_tests.log | tail -2
tools/ab_run.sh "--steps 20 --warmup 5" r5static r5tickets2 r5tickets3 > $O/ab_first_ticket_c2_20.log 2>&1; cat $O/ab_first_ticket_c2_20.log
tools/ab_run.sh "--config c3 --steps 20 --warmup 5" r5static r5tickets2 r5tickets3 > $O/ab_first_ticket_c3.log 2>&1; cat $O/ab_first_ticket_c3.log
tools/ab_run.sh "--steps 1 --warmup 3 --reps 9" r5static r5tickets2 r5tickets3 > $O/ab_first_ticket_tick1.log 2>&1; cat $O/ab_first_ticket_tick1.log

######## session_r05_24.sh
O=gpurun_out/r05; mkdir -p $O
tools/ab_run.sh "--steps 20 --warmup 5" r5t_base r5t_grid6 r5t_grid7 r5t_over1 > $O/ab_grid_sizes_c2_20.log 2>&1; cat $O/ab_grid_sizes_c2_20.log
tools/ab_run.sh "--config c3 --steps 20 --warmup 5" r5t_base r5t_grid6 r5t_grid7 r5t_over1 > $O/ab_grid_sizes_c3.log 2>&1; cat $O/ab_grid_sizes_c3.log
tools/ab_run.sh "--steps 1 --warmup 3 --reps 9" r5t_base r5t_grid6 r5t_over1 > $O/ab_grid_sizes_tick1.log 2>&1; cat $O/ab_grid_sizes_tick1.log

######## session_r05_25.sh
O=gpurun_out/r05; mkdir -p $O
tools/ab_run.sh "--steps 20 --warmup 5" r5u8 r5u4 r5u2 r5u16 > $O/ab_logic_u_tickets_c2_20.log 2>&1; cat $O/ab_logic_u_tickets_c2_20.log
tools/ab_run.sh "--config c3 --steps 20 --warmup 5" r5u8 r5u4 r5u2 > $O/ab_logic_u_tickets_c3.log 2>&1; cat $O/ab_logic_u_tickets_c3.log

######## session_r05_26.sh
O=gpurun_out/r05; mkdir -p $O
timeout 1200 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -p no:cacheprovider -k "stream or render_fused or fuzz or memory_limit or batch_is_halved or viewport" > $O/gpu_stream_tickets_tests.log 2>&1; grep -E "passed|failed|rror" $O/gpu_stream_tickets_tests.log | tail -2
tools/ab_run.sh "--steps 20 --warmup 5 --pipeline stream" r5st_static r5st_tickets > $O/ab_stream_tickets_c2_20.log 2>&1; cat $O/ab_stream_tickets_c2_20.log
tools/ab_run.sh "--steps 128 --warmup 128 --pipeline stream" r5st_static r5st_tickets > $O/ab_stream_tickets_c2_128.log 2>&1; cat $O/ab_stream_tickets_c2_128.log
tools/ab_run.sh "--config c3 --steps 20 --warmup 5 --pipeline stream" r5st_static r5st_tickets > $O/ab_stream_tickets_c3.log 2>&1; cat $O/ab_stream_tickets_c3.log
tools/ab_run.sh "--steps 20 --warmup 5" r5st_static r5st_tickets > $O/ab_stream_tickets_batch_c2_20.log 2>&1; cat $O/ab_stream_tickets_batch_c2_20.log

######## session_r05_27.sh
O=gpurun_out/r05; mkdir -p $O
tools/ab_run.sh "--steps 20 --warmup 5" r5lt512 r5lt256 r5lt256u16 > $O/ab_logic_threads_c2_20.log 2>&1; cat $O/ab_logic_threads_c2_20.log
tools/ab_run.sh "--config c3 --steps 20 --warmup 5" r5lt512 r5lt256 r5lt256u16 > $O/ab_logic_threads_c3.log 2>&1; cat $O/ab_logic_threads_c3.log

######## session_r05_28.sh
O=gpurun_out/r05; mkdir -p $O
tools/ab_args.sh "--config c3 --steps 20 --warmup 5" "--primary-form 0" "--primary-form 1" "--primary-form 2" > $O/ab_primary_form_tickets_c3.log 2>&1; cat $O/ab_primary_form_tickets_c3.log
tools/ab_args.sh "--steps 20 --warmup 5" "--primary-form 0" "--primary-form 1" "--primary-form 2" > $O/ab_primary_form_tickets_c2_20.log 2>&1; cat $O/ab_primary_form_tickets_c2_20.log
timeout 300 python3 bench.py --config c3 --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs --no-l1-microbench 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): print(json.loads(l)['config'].get('primary_form'))"

######## session_r05_29.sh
O=gpurun_out/r05; mkdir -p $O
V=("--trace-budget 48" "--trace-budget 24" "--trace-budget 96" "--trace-budget 192" "--trace-budget 0")
tools/ab_args.sh "--config c3 --steps 20 --warmup 5" "${V[@]}" > $O/scan_trace_budget_tickets_c3.log 2>&1; cat $O/scan_trace_budget_tickets_c3.log
tools/ab_args.sh "--steps 20 --warmup 5" "${V[@]}" > $O/scan_trace_budget_tickets_c2_20.log 2>&1; cat $O/scan_trace_budget_tickets_c2_20.log

######## session_r05_30.sh
O=gpurun_out/r05; mkdir -p $O
V=("--trace-budget 24" "--trace-budget 16" "--trace-budget 12" "--trace-budget 8")
tools/ab_args.sh "--steps 20 --warmup 5" "${V[@]}" > $O/scan_trace_budget_low_c2_20.log 2>&1; cat $O/scan_trace_budget_low_c2_20.log
tools/ab_args.sh "--config c3 --steps 20 --warmup 5" "${V[@]}" > $O/scan_trace_budget_low_c3.log 2>&1; cat $O/scan_trace_budget_low_c3.log

######## session_r05_31.sh
O=gpurun_out/r05; mkdir -p $O
tools/ab_run.sh "--steps 20 --warmup 5" r5g8 r5gres > $O/ab_trace_grid_resident_c2_20.log 2>&1; cat $O/ab_trace_grid_resident_c2_20.log
tools/ab_run.sh "--config c3 --steps 20 --warmup 5" r5g8 r5gres > $O/ab_trace_grid_resident_c3.log 2>&1; cat $O/ab_trace_grid_resident_c3.log
tools/ab_run.sh "--steps 128 --warmup 128" r5g8 r5gres > $O/ab_trace_grid_resident_c2_128.log 2>&1; cat $O/ab_trace_grid_resident_c2_128.log

######## session_r05_prof.sh
# usage (GPU box): bash tools/sessions/session_r05_prof.sh
# the committed evidence of the round: kernel trace + PMC passes of the timed configuration (c2) and of the two extra
# configs (c3, c5), the default bench lines, every configuration of DESIGN 7 on one box
O=gpurun_out/r05; mkdir -p $O
bash tools/prof_session.sh r05/final > $O/final_prof.log 2>&1; tail -2 $O/final_prof.log
bash tools/prof_session.sh r05/c3 --config c3 > $O/c3_prof.log 2>&1; tail -2 $O/c3_prof.log
bash tools/prof_session.sh r05/c5 --config c5 > $O/c5_prof.log 2>&1; tail -2 $O/c5_prof.log
timeout 600 python3 bench.py --steps 20 --warmup 5 > $O/final_bench_steps20.json.log 2>&1
timeout 900 python3 bench.py > $O/final_bench_default.json.log 2>&1
{ for a in "--steps 20 --warmup 5" "--steps 128 --warmup 128" "--steps 1 --warmup 3 --reps 9" "--steps 20 --warmup 5 --config c3" "--steps 128 --warmup 128 --config c3" "--steps 20 --warmup 5 --config c5" "--steps 20 --warmup 5 --textured" "--steps 20 --warmup 5 --width 3840 --height 2160" "--steps 20 --warmup 5 --pipeline stream" "--steps 4 --warmup 2 --pipeline megakernel"; do
  echo "== bench.py $a"
  timeout 900 python3 bench.py $a --no-cpu-baseline --no-extra-configs 2>/dev/null | python3 -c "
import sys,json
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); k=d['roofline'].get('kernels',{})
        print(d['value'], 'Msamples/s', d['ms_per_step'], 'ms/step', {c:v['ms_per_step'] for c,v in k.items()}, 'parity', (d.get('parity_check') or {}).get('equal'))
"
done; } > $O/final_configs_one_box.log 2>&1
cat $O/final_configs_one_box.log

