# FETCH_SIZE of the trace kernel on the 1 M-triangle scene for two node layouts (A/B libraries in ab_libs/)
cd ${GRAFT_REPO_ROOT:-/root/repo}; export TMPDIR=/tmp
for n in "$@"; do
  FSPT_LIB=$PWD/ab_libs/$n.so timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/c3f_$n -- python3 bench.py --config c3 --steps 32 --batch 32 --warmup 0 --reps 1 --no-cpu-baseline > gpurun_out/c3f_$n.log 2>&1
done
echo c3 layout done
