# (A/B of round 4, not adopted: pre-decoded flat-colour material sets; result: profiles/r04/README.md)
# A/B: material texture sets of flat colours pre-decoded in the table (cdec) against decoding per shading event (base)
O=gpurun_out/r04; mkdir -p $O
{
for cfg in "--steps 20 --warmup 5" "--steps 20 --warmup 5 --config c3" "--steps 128 --warmup 128"; do
  echo "#### $cfg"
  bash tools/ab_run.sh "$cfg" base cdec
done
} > $O/s17_const_sets.log 2>&1
cat $O/s17_const_sets.log
