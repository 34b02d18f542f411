# usage (GPU box): bash tools/sessions/session_r06_3.sh
# after the binning A/B (lost: session_r06_2): its counters for the record; then three small changes, each against the
# tree without it, one box, interleaved: the wave index as a scalar (no scratch in k_wf_trace / k_wf_tail), the resolve that
# reads whole rows, the environment map as bilinear footprint records; and the GPU suite on the tree's own library
O=gpurun_out/r06; mkdir -p $O
FSPT_LIB=$PWD/ab_libs/bin_all.so bash tools/pmc_rounds.sh r06/rounds_c2_bins > /dev/null 2>&1; tail -30 $O/rounds_c2_bins/rounds.txt | grep -E "kernel|trace|logic  *[12] "
FSPT_LIB=$PWD/ab_libs/bin_all.so bash tools/pmc_rounds.sh r06/rounds_c3_bins --config c3 > /dev/null 2>&1; grep -E "trace" $O/rounds_c3_bins/rounds.txt
timeout 1200 python3 -m pytest tests -x -q -m gpu > $O/gpu_suite_1.log 2>&1; tail -3 $O/gpu_suite_1.log
bash tools/ab_run.sh "--steps 20 --warmup 5" cur wu0 res1 envfp > $O/ab_small3_c2_20.log 2>&1; cat $O/ab_small3_c2_20.log
bash tools/ab_run.sh "--steps 20 --warmup 5 --config c3" cur wu0 res1 envfp > $O/ab_small3_c3.log 2>&1; cat $O/ab_small3_c3.log
bash tools/ab_run.sh "--steps 128 --warmup 128 --reps 3" cur wu0 res1 envfp > $O/ab_small3_c2_128.log 2>&1; cat $O/ab_small3_c2_128.log
bash tools/ab_run.sh "--steps 1 --warmup 3 --reps 9" cur wu0 > $O/ab_small3_tick1.log 2>&1; cat $O/ab_small3_tick1.log
