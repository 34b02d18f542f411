set -x
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04/s1_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r04/s1_pytest.log
tail -5 gpurun_out/r04/s1_pytest.log
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r04/s1_bench20.log 2>&1; tail -c 600 gpurun_out/r04/s1_bench20.log
tools/ab_run.sh "--steps 20 --warmup 5" base t256w4 t256w5 t512w6 > gpurun_out/r04/s1_ab_primary_occ.log 2>&1; cat gpurun_out/r04/s1_ab_primary_occ.log
tools/ab_run.sh "--steps 20 --warmup 5 --config c3" base > gpurun_out/r04/s1_c3.log 2>&1; cat gpurun_out/r04/s1_c3.log
tools/ab_run.sh "--steps 1 --warmup 3" base > gpurun_out/r04/s1_tick1.log 2>&1; cat gpurun_out/r04/s1_tick1.log
