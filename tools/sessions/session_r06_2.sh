# usage (GPU box): bash tools/sessions/session_r06_2.sh
# survivors written in bins (WF_BIN_PRIMARY / WF_BIN_LOGIC_PRE / WF_BIN_LOGIC_POST, fspt_kernels.hip):
# bit-equality of every variant (the bench's own parity check + a slice of the GPU suite on the all-bins build), then the
# A/B on one box, interleaved: C2 20 ticks, 1 M triangles, C2 128 ticks
O=gpurun_out/r06; mkdir -p $O
FSPT_LIB=$PWD/ab_libs/bin_all.so timeout 900 python3 -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "render or trace or fuzz or tail or stream or refract or textured or baseline or bench_configuration" > $O/gpu_bin_all_tests.log 2>&1; tail -3 $O/gpu_bin_all_tests.log
bash tools/ab_run.sh "--steps 20 --warmup 5" base bin_p bin_lpre bin_lpp bin_all > $O/ab_bins_c2_20.log 2>&1; cat $O/ab_bins_c2_20.log
bash tools/ab_run.sh "--steps 20 --warmup 5 --config c3" base bin_p bin_lpre bin_lpp bin_all > $O/ab_bins_c3.log 2>&1; cat $O/ab_bins_c3.log
bash tools/ab_run.sh "--steps 128 --warmup 128 --reps 3" base bin_all > $O/ab_bins_c2_128.log 2>&1; cat $O/ab_bins_c2_128.log
