# usage (GPU box): bash tools/sessions/session_r06_1.sh
# round 6, before anything changes: the driver's bench form on round 5's tree, and the per-round cache counters of the
# trace / logic launches on C2 and C3 ("measure first")
O=gpurun_out/r06; mkdir -p $O
timeout 600 python3 bench.py --steps 20 --warmup 5 --no-extra-configs > $O/head_bench_steps20.json.log 2>&1; tail -c 600 $O/head_bench_steps20.json.log; echo
bash tools/pmc_rounds.sh r06/rounds_c2_before
bash tools/pmc_rounds.sh r06/rounds_c3_before --config c3
