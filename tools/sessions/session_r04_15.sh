# (experiment of round 4 that was NOT adopted: the sources it ran - primary form 3 / the tail kernel with an LDS copy of the top of the tree - are not in the tree; results: profiles/r04/README.md)
O=gpurun_out/r04; mkdir -p $O
{
for cfg in "--steps 1 --warmup 3 --batch 1" "--steps 20 --warmup 5 --config c3" "--steps 20 --warmup 5"; do
  echo "#### $cfg"
  bash tools/ab_run.sh "$cfg" tail0 tail256 bfs1k tail1k
done
} > $O/s15_tail_lds_top.log 2>&1
cat $O/s15_tail_lds_top.log
