cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g2
timeout 900 python -m pytest tests -m gpu -x -q -k "stream or fused or refractive or two_call or ragged" > gpurun_out/g2/pytest_stream.log 2>&1; echo "rc=$?" >> gpurun_out/g2/pytest_stream.log
tail -15 gpurun_out/g2/pytest_stream.log
for cfg in "wavefront 0" "stream 2097152" "stream 4194304" "stream 8388608" "stream 16777216" "stream 33554432" "stream2 4194304" "stream2 8388608"; do
  set -- $cfg
  for steps in "20 5" "128 128"; do
    set -- $cfg; pl=$1; pool=$2; set -- $steps
    echo "== $pl pool=$pool steps=$1" >> gpurun_out/g2/sweep.log
    timeout 300 python bench.py --pipeline $pl --pool $pool --steps $1 --warmup $2 --no-cpu-baseline 2>>gpurun_out/g2/sweep.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l)
        k = j['roofline'].get('kernels', {})
        print(j['value'], j['ms_per_step'], j.get('parity_check'), {a: (b['ms_per_step'], b['launches']) for a, b in k.items()})
" >> gpurun_out/g2/sweep.log 2>&1
  done
done
cat gpurun_out/g2/sweep.log
tools/fetch_calib.sh g2/calib > gpurun_out/g2/calib.log 2>&1
tail -26 gpurun_out/g2/calib.log
