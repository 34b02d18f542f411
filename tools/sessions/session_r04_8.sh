O=gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/s8_pytest.log 2>&1; tail -3 $O/s8_pytest.log
tools/ab_run.sh "--steps 20 --warmup 5" r1 r2 r2s4 r2s16 r3 > $O/ab_primary_refill.log 2>&1; cat $O/ab_primary_refill.log
tools/ab_run.sh "--steps 20 --warmup 5 --config c3" r1 r2 r2s16 r3 > $O/ab_primary_refill_c3.log 2>&1; cat $O/ab_primary_refill_c3.log
tools/ab_run.sh "--steps 1 --warmup 3" r1 r2 > $O/ab_primary_refill_tick1.log 2>&1; cat $O/ab_primary_refill_tick1.log
tools/ab_run.sh "--steps 128 --warmup 128" r1 r2 > $O/ab_primary_refill_128.log 2>&1; cat $O/ab_primary_refill_128.log
