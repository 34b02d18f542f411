# usage (GPU box): bash tools/sessions/session_r06_16.sh
# -fno-slp-vectorize (the SLP vectorizer builds 3-vectors out of the ray's components and takes them apart through SCRATCH,
# and costs the shading kernels ~15 registers), alone, with a fifth wave for the shading kernels, and with the hottest leaf
# records in the trace kernel's LDS: bit-equality first, then the A/B against the committed tree (h_head)
O=gpurun_out/r06; mkdir -p $O
FSPT_LIB=$PWD/ab_libs/n_slp0_hot2_w5.so timeout 1500 python3 -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "math or intersect or render or fuzz or counters or chain or baseline or suspended or tail or textured or refract or stream" > $O/gpu_noslp_tests.log 2>&1; tail -2 $O/gpu_noslp_tests.log
bash tools/ab_run.sh "--steps 20 --warmup 5" h_head n_slp0 n_slp0_w5 n_slp0_hot2 n_slp0_hot4 n_slp0_hot2_w5 > $O/ab_noslp_c2_20.log 2>&1; cat $O/ab_noslp_c2_20.log
bash tools/ab_run.sh "--steps 20 --warmup 5 --config c3" h_head n_slp0 n_slp0_w5 n_slp0_hot2 n_slp0_hot2_w5 > $O/ab_noslp_c3.log 2>&1; cat $O/ab_noslp_c3.log
bash tools/ab_run.sh "--steps 20 --warmup 5 --textured" h_head n_slp0 n_slp0_w5 > $O/ab_noslp_tex.log 2>&1; cat $O/ab_noslp_tex.log
