# soak on the final kernels: 1000 fuzz seeds (seeds 0-399 are profiles/r04/fuzz_soak_400_seeds.log's)
O=gpurun_out/r04; mkdir -p $O
FSPT_FUZZ_SEEDS=1000 timeout 3300 python3 -m pytest tests/test_parity_gpu.py -m gpu -q -k "fuzz_random_scenes" -p no:cacheprovider > $O/fuzz_soak_1000_seeds.log 2>&1; tail -3 $O/fuzz_soak_1000_seeds.log
