# (timing experiment of round 4, results discarded: shade_hit without its reflect branch; result: profiles/r04/README.md)
# TIMING EXPERIMENT (results discarded): shade_hit never takes its reflect branch (nospec) - the upper bound of what
# deferring the few specular lanes of a wave to a dense pass could save - against the real kernels (base)
O=gpurun_out/r04; mkdir -p $O
{
for cfg in "--steps 20 --warmup 5 --no-parity-check" "--steps 20 --warmup 5 --config c3 --no-parity-check"; do
  echo "#### $cfg"
  bash tools/ab_run.sh "$cfg" base nospec
done
} > $O/s27_exp_no_specular.log 2>&1
cat $O/s27_exp_no_specular.log
