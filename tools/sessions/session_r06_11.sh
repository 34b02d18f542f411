# usage (GPU box): bash tools/sessions/session_r06_11.sh
# the final tree after the constants scan: sessions 5 and 7 again (profiles for the stamp, GPU suite, smoke, default bench
# line, every configuration on one box, the Node host, a fuzz soak)
O=gpurun_out/r06c; mkdir -p $O
timeout 1500 python3 -m pytest tests -q -m gpu > $O/gpu_suite_final.log 2>&1; grep -E "passed|failed" $O/gpu_suite_final.log | tail -1
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
bash tools/prof_session.sh r06c/final > $O/final_prof.log 2>&1; tail -1 $O/final_prof.log
bash tools/prof_session.sh r06c/c3 --config c3 > $O/c3_prof.log 2>&1; tail -1 $O/c3_prof.log
bash tools/prof_session.sh r06c/c5 --config c5 > $O/c5_prof.log 2>&1; tail -1 $O/c5_prof.log
bash tools/prof_session.sh r06c/tex --textured > $O/tex_prof.log 2>&1; tail -1 $O/tex_prof.log
{ for a in "--steps 20 --warmup 5" "--steps 128 --warmup 128" "--steps 1 --warmup 3 --reps 9" "--tick-mode --steps 128 --warmup 8 --reps 3" "--steps 20 --warmup 5 --config c3" "--steps 128 --warmup 128 --config c3" "--steps 20 --warmup 5 --config c5" "--steps 20 --warmup 5 --textured" "--steps 20 --warmup 5 --width 3840 --height 2160" "--steps 20 --warmup 5 --pipeline stream" "--steps 4 --warmup 2 --pipeline megakernel"; do
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
python3 tools/write_bench_scene.py /tmp/benchscene > /dev/null 2>&1 && timeout 600 node fspt_amd/js/bench.js --scene /tmp/benchscene/scene/bench.json --focal-depth 2 --aperture 0.02 --steps 20 --warmup 5 > $O/bench_node_host.json.log 2>&1; python3 -c "
import json
for l in open('$O/bench_node_host.json.log'):
    if l.startswith('{'): d=json.loads(l); print('node host', d['value'])"
FSPT_FUZZ_SEEDS=400 timeout 2400 python3 -m pytest tests/test_parity_gpu.py -m gpu -q -k "fuzz_random_scenes" -p no:cacheprovider > $O/fuzz_soak_400_seeds_final.log 2>&1; grep -E "passed|failed" $O/fuzz_soak_400_seeds_final.log | tail -1
( time timeout 900 python3 bench.py --steps 20 --warmup 5 ) > $O/bench_steps20_unstamped.json.log 2> $O/bench_steps20_unstamped.time; tail -3 $O/bench_steps20_unstamped.time
