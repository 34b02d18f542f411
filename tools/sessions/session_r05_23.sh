O=gpurun_out/r05; mkdir -p $O
FSPT_LIB=$PWD/ab_libs/r5tickets3.so timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -p no:cacheprovider -k "render_fused or primary_launch or fuzz or two_call or ragged or viewport or stream_scheduler or suspended or refractive or tail_kernel" > $O/gpu_tickets3_tests.log 2>&1; grep -E "passed|failed|rror" $O/gpu_tickets3_tests.log | tail -2
tools/ab_run.sh "--steps 20 --warmup 5" r5static r5tickets2 r5tickets3 > $O/ab_first_ticket_c2_20.log 2>&1; cat $O/ab_first_ticket_c2_20.log
tools/ab_run.sh "--config c3 --steps 20 --warmup 5" r5static r5tickets2 r5tickets3 > $O/ab_first_ticket_c3.log 2>&1; cat $O/ab_first_ticket_c3.log
tools/ab_run.sh "--steps 1 --warmup 3 --reps 9" r5static r5tickets2 r5tickets3 > $O/ab_first_ticket_tick1.log 2>&1; cat $O/ab_first_ticket_tick1.log
