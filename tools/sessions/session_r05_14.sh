O=gpurun_out/r05; mkdir -p $O
FSPT_LIB=$PWD/ab_libs/r5xcd.so timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -p no:cacheprovider -k "render_fused or suspended or fuzz or two_call or refractive or full_size or million" > $O/gpu_xcd_tests.log 2>&1; grep -E "passed|failed" $O/gpu_xcd_tests.log | tail -1
tools/ab_run.sh "--steps 20 --warmup 5" r5base r5xcd > $O/ab_trace_xcd_c2_20.log 2>&1; cat $O/ab_trace_xcd_c2_20.log
tools/ab_run.sh "--steps 128 --warmup 128" r5base r5xcd > $O/ab_trace_xcd_c2_128.log 2>&1; cat $O/ab_trace_xcd_c2_128.log
tools/ab_run.sh "--config c3 --steps 20 --warmup 5" r5base r5xcd > $O/ab_trace_xcd_c3.log 2>&1; cat $O/ab_trace_xcd_c3.log
tools/ab_run.sh "--steps 20 --warmup 5 --width 3840 --height 2160" r5base r5xcd > $O/ab_trace_xcd_4k.log 2>&1; cat $O/ab_trace_xcd_4k.log
