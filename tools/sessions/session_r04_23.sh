# A/B: the six throughput divisions of a shading event once behind the reflect / Lambert branches (cdiv) instead of in both (base)
O=gpurun_out/r04; mkdir -p $O
{
FSPT_LIB=$PWD/ab_libs/cdiv.so timeout 900 python3 -m pytest tests/test_parity_gpu.py -q -x -k "fuzz or refractive or textured or bench_configuration or megakernel or brdf" 2>&1 | tail -2
for cfg in "--steps 20 --warmup 5" "--steps 20 --warmup 5 --config c3" "--steps 128 --warmup 128" "--steps 20 --warmup 5 --textured"; do
  echo "#### $cfg"
  bash tools/ab_run.sh "$cfg" base cdiv
done
} > $O/s23_common_divisions.log 2>&1
cat $O/s23_common_divisions.log
