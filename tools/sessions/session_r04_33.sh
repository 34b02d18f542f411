# A/B: the last trace launch in front of the tail kernel suspends its long rays too, the tail kernel traces them again
# from the root (lastsusp), against letting them finish in the trace launch (base)
O=gpurun_out/r04; mkdir -p $O
{
FSPT_LIB=$PWD/ab_libs/lastsusp.so timeout 900 python3 -m pytest tests/test_parity_gpu.py -q -x -k "fuzz or refractive or suspended or bench_configuration or million" 2>&1 | tail -2
for cfg in "--steps 20 --warmup 5" "--steps 20 --warmup 5 --config c3" "--steps 1 --warmup 3"; do
  echo "#### $cfg"
  bash tools/ab_run.sh "$cfg" base lastsusp
done
} > $O/s33_last_trace_suspends.log 2>&1
cat $O/s33_last_trace_suspends.log
