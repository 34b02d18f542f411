set -x
O=gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/s3_pytest.log 2>&1; tail -3 $O/s3_pytest.log
tools/ab_run.sh "--steps 1 --warmup 3" ts1e9 ts32 ts32p ts16 ts64 > $O/ab_tail_slice_tick1.log 2>&1; cat $O/ab_tail_slice_tick1.log
tools/ab_run.sh "--steps 20 --warmup 5 --config c3" ts1e9 ts32 ts32p ts16 ts64 > $O/ab_tail_slice_c3.log 2>&1; cat $O/ab_tail_slice_c3.log
tools/ab_run.sh "--steps 20 --warmup 5" ts1e9 ts32 ts32p ts16 ts64 > $O/ab_tail_slice.log 2>&1; cat $O/ab_tail_slice.log
