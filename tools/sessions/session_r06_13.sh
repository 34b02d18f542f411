# usage (GPU box): bash tools/sessions/session_r06_13.sh
# what the driver runs at round end, on the final tree with the re-stamped traffic: the default bench line (twice) and the Node bench test
O=gpurun_out/r06; mkdir -p $O
( time timeout 900 python3 bench.py --steps 20 --warmup 5 ) > $O/final_bench_steps20.json.log 2> $O/final_bench_steps20.time; tail -3 $O/final_bench_steps20.time
( time timeout 1200 python3 bench.py ) > $O/final_bench_default.json.log 2> $O/final_bench_default.time; tail -3 $O/final_bench_default.time
