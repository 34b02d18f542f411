# the three full bench lines once more, with profiles/hbm_traffic.json stamped for the sources they run
O=gpurun_out/r04; mkdir -p $O
timeout 600 python3 bench.py --steps 20 --warmup 5 > $O/final_bench_steps20.json.log 2>&1
timeout 600 python3 bench.py > $O/final_bench_default.json.log 2>&1
timeout 600 python3 bench.py --steps 20 --warmup 5 --config c3 > $O/c3_bench_steps20.json.log 2>&1
tail -c 200 $O/c3_bench_steps20.json.log
