O=gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "not full_size and not million and not 4k" > $O/s9_pytest.log 2>&1; tail -2 $O/s9_pytest.log
tools/ab_run.sh "--steps 20 --warmup 5" r1 r2b r2bs16 > $O/ab_primary_refill2.log 2>&1; cat $O/ab_primary_refill2.log
tools/ab_run.sh "--steps 20 --warmup 5 --config c3" r1 r2b r2bs16 > $O/ab_primary_refill2_c3.log 2>&1; cat $O/ab_primary_refill2_c3.log
tools/ab_run.sh "--steps 1 --warmup 3" r1 r2b > $O/ab_primary_refill2_tick1.log 2>&1; cat $O/ab_primary_refill2_tick1.log
tools/ab_run.sh "--steps 128 --warmup 128" r1 r2b > $O/ab_primary_refill2_128.log 2>&1; cat $O/ab_primary_refill2_128.log
