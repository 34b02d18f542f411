# usage (GPU box): bash tools/sessions/session_r06_14.sh
# assurance on the final tree: the GPU suite three times over (flakiness), a 1000-seed fuzz soak
O=gpurun_out/r06; mkdir -p $O
for i in 1 2 3; do timeout 1500 python3 -m pytest tests -q -m gpu -p no:cacheprovider > $O/gpu_suite_x3_$i.log 2>&1; grep -E "passed|failed" $O/gpu_suite_x3_$i.log | tail -1; done
FSPT_FUZZ_SEEDS=1000 timeout 3000 python3 -m pytest tests/test_parity_gpu.py -m gpu -q -k "fuzz_random_scenes" -p no:cacheprovider > $O/fuzz_soak_1000_seeds_final.log 2>&1; grep -E "passed|failed" $O/fuzz_soak_1000_seeds_final.log | tail -1
