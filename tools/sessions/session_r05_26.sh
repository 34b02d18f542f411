O=gpurun_out/r05; mkdir -p $O
timeout 1200 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -p no:cacheprovider -k "stream or render_fused or fuzz or memory_limit or batch_is_halved or viewport" > $O/gpu_stream_tickets_tests.log 2>&1; grep -E "passed|failed|rror" $O/gpu_stream_tickets_tests.log | tail -2
tools/ab_run.sh "--steps 20 --warmup 5 --pipeline stream" r5st_static r5st_tickets > $O/ab_stream_tickets_c2_20.log 2>&1; cat $O/ab_stream_tickets_c2_20.log
tools/ab_run.sh "--steps 128 --warmup 128 --pipeline stream" r5st_static r5st_tickets > $O/ab_stream_tickets_c2_128.log 2>&1; cat $O/ab_stream_tickets_c2_128.log
tools/ab_run.sh "--config c3 --steps 20 --warmup 5 --pipeline stream" r5st_static r5st_tickets > $O/ab_stream_tickets_c3.log 2>&1; cat $O/ab_stream_tickets_c3.log
tools/ab_run.sh "--steps 20 --warmup 5" r5st_static r5st_tickets > $O/ab_stream_tickets_batch_c2_20.log 2>&1; cat $O/ab_stream_tickets_batch_c2_20.log
