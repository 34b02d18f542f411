cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g10
timeout 900 python -m pytest tests -m gpu -x -q --timeout=120 --timeout-method=thread > gpurun_out/g10/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/g10/pytest.log
tail -4 gpurun_out/g10/pytest.log
run() { echo "== $*" >> gpurun_out/g10/cfg.log; timeout 300 python bench.py --no-cpu-baseline "$@" 2>>gpurun_out/g10/cfg.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l)
        k = j['roofline'].get('kernels', {})
        print(j['value'], j['ms_per_step'], (j.get('parity_check') or {}).get('equal'), j['roofline'].get('frac'), {a: (b['ms_per_step'], b['launches']) for a, b in k.items()})
" >> gpurun_out/g10/cfg.log 2>&1; }
run --steps 20 --warmup 5
run
run --config c3
run --config c3 --steps 20 --warmup 5
run --config c5
run --textured
run --width 3840 --height 2160 --scaling strong
run --pipeline megakernel --steps 16 --warmup 4
run --pipeline stream --steps 20 --warmup 5
run --pipeline stream
run --pipeline stream --pool 4194304
run --steps 1 --warmup 3 --reps 9
cat gpurun_out/g10/cfg.log
