cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g3
timeout 900 python -m pytest tests -m gpu -x -q -k "stream or fused or refractive or two_call or ragged or bounces" > gpurun_out/g3/pytest_stream.log 2>&1; echo "rc=$?" >> gpurun_out/g3/pytest_stream.log
tail -5 gpurun_out/g3/pytest_stream.log
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for cfg in "wavefront 0" "stream 16777216" "stream 4194304"; do
  set -- $cfg
  rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/g3/kt_$1_$2 -- python3 $R/bench.py --pipeline $1 --pool $2 --steps 20 --warmup 5 --no-cpu-baseline --no-parity-check > $R/gpurun_out/g3/kt_$1_$2.log 2>&1
  echo "== $1 $2"; python3 $R/tools/timeline.py $R/gpurun_out/g3/kt_$1_$2 | tail -6
  find $R/gpurun_out/g3/kt_$1_$2 -name "*.csv" -size +3M -delete
done
cd $R
python bench.py --pipeline stream --pool 2097152 --no-cpu-baseline | cut -c1-600
