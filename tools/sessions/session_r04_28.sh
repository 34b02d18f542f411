# (A/B of round 4, not adopted: reflecting lanes of a shading wave parked for a dense launch; result: profiles/r04/README.md)
# A/B: a wave's few reflecting lanes parked for k_wf_spec (defer) against shaded in place (nodefer)
O=gpurun_out/r04; mkdir -p $O
{
FSPT_LIB=$PWD/ab_libs/defer.so timeout 1500 python3 -m pytest tests/test_parity_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3
for cfg in "--steps 20 --warmup 5" "--steps 20 --warmup 5 --config c3" "--steps 128 --warmup 128"; do
  echo "#### $cfg"
  bash tools/ab_run.sh "$cfg" nodefer defer
done
} > $O/s28_defer_reflect.log 2>&1
cat $O/s28_defer_reflect.log
