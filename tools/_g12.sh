cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g12
timeout 900 python -m pytest tests -m gpu -x -q --timeout=120 --timeout-method=thread > gpurun_out/g12/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/g12/pytest.log
tail -3 gpurun_out/g12/pytest.log
grep -q "rc=0" gpurun_out/g12/pytest.log || exit 0
run() { echo "== $*" >> gpurun_out/g12/ab.log; timeout 300 python bench.py --no-cpu-baseline "$@" 2>>gpurun_out/g12/ab.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l)
        k = j['roofline'].get('kernels', {})
        print(j['value'], j['ms_per_step'], (j.get('parity_check') or {}).get('equal'), {a: (b['ms_per_step'], b['launches']) for a, b in k.items()})
" >> gpurun_out/g12/ab.log 2>&1; }
for rep in 1 2 3; do
for b in 0 24; do
  run --steps 20 --warmup 5 --trace-budget $b
done
done
for b in 0 24; do
  run --steps 128 --warmup 128 --trace-budget $b
  run --steps 1 --warmup 3 --reps 9 --trace-budget $b
  run --pipeline stream --pool 8388608 --steps 128 --warmup 128 --trace-budget $b
  run --pipeline stream --steps 20 --warmup 5 --trace-budget $b
done
run --bounces 1 --steps 20 --warmup 5
cat gpurun_out/g12/ab.log
