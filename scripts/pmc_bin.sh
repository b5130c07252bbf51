cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for set in "VALUBusy SALUBusy" "MemUnitBusy MemUnitStalled" "WriteUnitStalled LDSBankConflict" "OccupancyPercent" "SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
  tag=$(echo $set | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmc_bin/$tag -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-aux > /dev/null 2> gpurun_out/pmc_bin/$tag.err || echo "fail $tag"
done
find gpurun_out/pmc_bin -name "*counter_collection.csv" | head -20
