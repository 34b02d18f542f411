O=gpurun_out/r05; mkdir -p $O
tools/ab_run.sh "--steps 20 --warmup 5" r5u8 r5u4 r5u2 r5u16 > $O/ab_logic_u_tickets_c2_20.log 2>&1; cat $O/ab_logic_u_tickets_c2_20.log
tools/ab_run.sh "--config c3 --steps 20 --warmup 5" r5u8 r5u4 r5u2 > $O/ab_logic_u_tickets_c3.log 2>&1; cat $O/ab_logic_u_tickets_c3.log
