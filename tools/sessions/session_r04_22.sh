# (A/B of round 4, not adopted: the logic kernel with its finishing paths processed densely; result: profiles/r04/README.md)
# A/B: k_wf_logic with the finishing paths listed and processed densely too (densefin) against every thread finishing its own (base)
O=gpurun_out/r04; mkdir -p $O
{
FSPT_LIB=$PWD/ab_libs/densefin.so timeout 900 python3 -m pytest tests/test_parity_gpu.py -q -x -k "fuzz or refractive or textured or counters or suspended or bench_configuration" 2>&1 | tail -2
for cfg in "--steps 20 --warmup 5" "--steps 20 --warmup 5 --config c3" "--steps 128 --warmup 128" "--steps 20 --warmup 5 --textured"; do
  echo "#### $cfg"
  bash tools/ab_run.sh "$cfg" base densefin
done
} > $O/s22_logic_dense_fin.log 2>&1
cat $O/s22_logic_dense_fin.log
