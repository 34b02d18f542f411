O=gpurun_out/r05; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=20 > $O/gpu_suite_durations.log 2>&1; grep -E "passed|failed|s call|s setup" $O/gpu_suite_durations.log | tail -24
for i in 1 2 3; do timeout 2400 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/gpu_suite_x3_$i.log 2>&1; grep -E "passed|failed" $O/gpu_suite_x3_$i.log | tail -1; done
FSPT_FUZZ_SEEDS=1000 timeout 3000 python -m pytest tests/test_parity_gpu.py -m gpu -q -k "fuzz_random_scenes" -p no:cacheprovider > $O/fuzz_soak_1000_seeds.log 2>&1; grep -E "passed|failed" $O/fuzz_soak_1000_seeds.log | tail -1
