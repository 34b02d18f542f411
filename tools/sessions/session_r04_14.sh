# (experiment of round 4 that was NOT adopted: the sources it ran - primary form 3 / the tail kernel with an LDS copy of the top of the tree - are not in the tree; results: profiles/r04/README.md)
O=$GRAFT_REPO_ROOT/gpurun_out/r04; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
i=0
for set in \
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU" \
 "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM" \
 "TA_TA_BUSY_sum TD_TD_BUSY_sum GRBM_GUI_ACTIVE" \
 "SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_MISC SQ_WAVES SQ_INSTS_SMEM"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --output-format csv -d /tmp/split_sq$i -- python3 $R/bench.py --steps 20 --warmup 0 --reps 1 --primary-form 3 --no-cpu-baseline --no-parity-check --no-l1-microbench > /tmp/split_sq$i.log 2>&1
  tail -2 /tmp/split_sq$i.log | cut -c1-200
done
python3 $R/tools/pmc_quick.py /tmp/split_sq1 /tmp/split_sq2 /tmp/split_sq3 /tmp/split_sq4 > $O/s14_split_pmc.txt 2>&1
cat $O/s14_split_pmc.txt
