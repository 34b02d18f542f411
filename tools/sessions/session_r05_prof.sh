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
