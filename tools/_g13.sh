cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g13
timeout 600 python -m pytest tests -m gpu -x -q --timeout=120 --timeout-method=thread -k "finish_kernel or suspended or stream_scheduler or tail_kernel" > gpurun_out/g13/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/g13/pytest.log
tail -3 gpurun_out/g13/pytest.log
grep -q "rc=0" gpurun_out/g13/pytest.log || exit 0
run() { echo "== $*" >> gpurun_out/g13/ab.log; timeout 300 python bench.py --no-cpu-baseline "$@" 2>>gpurun_out/g13/ab.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l)
        k = j['roofline'].get('kernels', {})
        print(j['value'], j['ms_per_step'], (j.get('parity_check') or {}).get('equal'), {a: (b['ms_per_step'], b['launches']) for a, b in k.items()})
" >> gpurun_out/g13/ab.log 2>&1; }
for rep in 1 2; do
for m in 0 1 2; do
  run --steps 20 --warmup 5 --finish-kernel $m
done
done
for m in 0 1 2; do
  run --steps 128 --warmup 128 --finish-kernel $m
done
for m in 0 1; do
  run --pipeline stream --steps 128 --warmup 128 --finish-kernel $m
  run --textured --steps 20 --warmup 5 --finish-kernel $m
done
cat gpurun_out/g13/ab.log
