O=gpurun_out/r04; mkdir -p $O
FSPT_FUZZ_SEEDS=400 timeout 2400 python -m pytest tests/test_parity_gpu.py -m gpu -q -k "fuzz_random_scenes" -p no:cacheprovider > $O/fuzz_soak_400_seeds.log 2>&1; tail -3 $O/fuzz_soak_400_seeds.log
