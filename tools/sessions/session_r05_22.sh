O=gpurun_out/r05; mkdir -p $O
FSPT_LIB=$PWD/ab_libs/r5tickets2.so timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -p no:cacheprovider -k "render_fused or primary_launch or fuzz or two_call or ragged or viewport or stream_scheduler or suspended or refractive or tail_kernel" > $O/gpu_tickets2_tests.log 2>&1; grep -E "passed|failed|rror" $O/gpu_tickets2_tests.log | tail -2
tools/ab_run.sh "--steps 20 --warmup 5" r5static r5tickets r5tickets2 > $O/ab_logic_tickets_c2_20.log 2>&1; cat $O/ab_logic_tickets_c2_20.log
tools/ab_run.sh "--config c3 --steps 20 --warmup 5" r5static r5tickets r5tickets2 > $O/ab_logic_tickets_c3.log 2>&1; cat $O/ab_logic_tickets_c3.log
tools/ab_run.sh "--steps 128 --warmup 128" r5tickets r5tickets2 > $O/ab_logic_tickets_c2_128.log 2>&1; cat $O/ab_logic_tickets_c2_128.log
