O=gpurun_out/r04; mkdir -p $O
timeout 1200 python3 -m pytest tests/test_parity_gpu.py -q -x -k "two_ranks_share" > $O/s20_share_gpu.log 2>&1; tail -15 $O/s20_share_gpu.log
timeout 600 python3 bench.py --gpus 2 --share-gpu --steps 20 --warmup 5 --no-l1-microbench > $O/s20_share_gpu_bench.log 2>&1; tail -c 1500 $O/s20_share_gpu_bench.log
