cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g8
timeout 1200 python -m pytest tests -m gpu -x -q -k "stream_path_state or suspended" > gpurun_out/g8/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/g8/pytest.log
tail -4 gpurun_out/g8/pytest.log
true
run() { echo "== $*" >> gpurun_out/g8/ab.log; timeout 300 python bench.py --no-cpu-baseline "$@" 2>>gpurun_out/g8/ab.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l)
        k = j['roofline'].get('kernels', {})
        print(j['value'], j['ms_per_step'], (j.get('parity_check') or {}).get('equal'), {a: (b['ms_per_step'], b['launches']) for a, b in k.items()})
" >> gpurun_out/g8/ab.log 2>&1; }
for rep in 1 2; do
for b in 0 16 24 32 48; do
  run --steps 20 --warmup 5 --trace-budget $b
done
done
for t in 4 6 7; do run --steps 20 --warmup 5 --trace-budget 24 --tail $t; done
for b in 0 24; do
  run --steps 128 --warmup 128 --trace-budget $b
  run --steps 1 --warmup 3 --reps 9 --trace-budget $b
  run --pipeline stream --pool 8388608 --steps 128 --warmup 128 --trace-budget $b
  run --pipeline stream --pool 8388608 --steps 20 --warmup 5 --trace-budget $b
  run --pipeline stream --pool 16777216 --steps 20 --warmup 5 --trace-budget $b
  run --pipeline stream --pool 4194304 --steps 128 --warmup 128 --trace-budget $b
done
cat gpurun_out/g8/ab.log
