O=gpurun_out/r05; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "bench or share_the_gpu or eight_ranks or rccl or js_" > $O/gpu_bench_tests.log 2>&1; grep -E "passed|failed" $O/gpu_bench_tests.log | tail -2
bash tools/sessions/session_r05_prof.sh
