O=gpurun_out/r05; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "rccl or pack_and_unpack or close_executes or tile_gather or share_the_gpu or multi_device or two_level" > $O/gpu_new_tests.log 2>&1; tail -5 $O/gpu_new_tests.log
tools/ab_run.sh "--steps 20 --warmup 5" r5c0 r5c32 r5c64 r5sinf32 > $O/ab_carry_sin_c2_20.log 2>&1; cat $O/ab_carry_sin_c2_20.log
tools/ab_run.sh "--config c3 --steps 20 --warmup 5" r5c0 r5c32 r5sinf32 > $O/ab_carry_sin_c3.log 2>&1; cat $O/ab_carry_sin_c3.log
