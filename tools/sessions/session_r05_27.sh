O=gpurun_out/r05; mkdir -p $O
tools/ab_run.sh "--steps 20 --warmup 5" r5lt512 r5lt256 r5lt256u16 > $O/ab_logic_threads_c2_20.log 2>&1; cat $O/ab_logic_threads_c2_20.log
tools/ab_run.sh "--config c3 --steps 20 --warmup 5" r5lt512 r5lt256 r5lt256u16 > $O/ab_logic_threads_c3.log 2>&1; cat $O/ab_logic_threads_c3.log
