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
