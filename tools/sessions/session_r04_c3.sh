O=gpurun_out/r04; mkdir -p $O
bash tools/prof_r04.sh r04/c3 --config c3 > $O/c3_prof.log 2>&1; tail -1 $O/c3_prof.log
timeout 600 python3 bench.py --steps 20 --warmup 5 --config c3 > $O/c3_bench_steps20.json.log 2>&1; tail -c 300 $O/c3_bench_steps20.json.log
