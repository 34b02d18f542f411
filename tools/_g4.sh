cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g4
timeout 900 python -m pytest tests -m gpu -x -q -k "stream" > gpurun_out/g4/pytest_stream.log 2>&1; echo "rc=$?" >> gpurun_out/g4/pytest_stream.log
tail -4 gpurun_out/g4/pytest_stream.log
for cfg in "stream 4194304 0" "stream 8388608 0" "stream 16777216 0" "stream 33554432 0" "stream 8388608 1"; do
  set -- $cfg
  for steps in "20 5" "128 128"; do
    set -- $cfg; pl=$1; pool=$2; ov=$3; set -- $steps
    echo "== $pl pool=$pool overlap=$ov steps=$1" >> gpurun_out/g4/sweep.log
    timeout 300 python bench.py --pipeline $pl --pool $pool --overlap $ov --steps $1 --warmup $2 --no-cpu-baseline 2>>gpurun_out/g4/sweep.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l)
        k = j['roofline'].get('kernels', {})
        print(j['value'], j['ms_per_step'], j.get('parity_check', {}).get('equal'), {a: (b['ms_per_step'], b['launches']) for a, b in k.items()})
" >> gpurun_out/g4/sweep.log 2>&1
  done
done
cat gpurun_out/g4/sweep.log
for d in 0 1 2 3; do echo "== drain $d"; python bench.py --pipeline stream --pool 16777216 --overlap 0 --drain $d --steps 20 --warmup 5 --no-cpu-baseline --no-parity-check | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(j['value'], j['ms_per_step'])
"; done
