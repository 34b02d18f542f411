O=gpurun_out/r05; mkdir -p $O
tools/ab_run.sh "--steps 20 --warmup 5" r5base r5envnt r5hitnt > $O/ab_nontemporal_c2_20.log 2>&1; cat $O/ab_nontemporal_c2_20.log
tools/ab_run.sh "--config c3 --steps 20 --warmup 5" r5base r5envnt r5hitnt > $O/ab_nontemporal_c3.log 2>&1; cat $O/ab_nontemporal_c3.log
