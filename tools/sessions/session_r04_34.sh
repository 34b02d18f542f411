# soak after the last change (the last trace launch suspends too): 200 fuzz seeds
O=gpurun_out/r04; mkdir -p $O
FSPT_FUZZ_SEEDS=200 timeout 1200 python3 -m pytest tests/test_parity_gpu.py -m gpu -q -k "fuzz_random_scenes" -p no:cacheprovider > $O/fuzz_soak_200_seeds_final.log 2>&1; tail -2 $O/fuzz_soak_200_seeds_final.log
