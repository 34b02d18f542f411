set -x
O=gpurun_out/r04; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q -k "not full_size and not million and not 4k" > $O/s2_pytest.log 2>&1; tail -3 $O/s2_pytest.log
tools/ab_run.sh "--steps 20 --warmup 5" base p256t p512t p256b p128t p256tu > $O/ab_primary_ticket.log 2>&1; cat $O/ab_primary_ticket.log
tools/ab_run.sh "--steps 20 --warmup 5 --config c3" base p256t p256tu > $O/ab_primary_ticket_c3.log 2>&1; cat $O/ab_primary_ticket_c3.log
tools/ab_run.sh "--steps 1 --warmup 3" base p256t p256tu > $O/ab_primary_ticket_tick1.log 2>&1; cat $O/ab_primary_ticket_tick1.log
tools/ab_run.sh "--steps 128 --warmup 128" base p256t p256tu > $O/ab_primary_ticket_128.log 2>&1; cat $O/ab_primary_ticket_128.log
