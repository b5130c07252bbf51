#!/bin/bash
# PMC counters for the kernels of one bench step: bash scripts/pmc_kernel.sh <outdir>; post-process with scripts/pmc_summary.py
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=${1:-gpurun_out/pmc_k}
mkdir -p $out
for set in "VALUBusy SALUBusy" "MemUnitBusy MemUnitStalled" "WriteUnitStalled LDSBankConflict" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY" "SQ_WAIT_ANY SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD" "TA_BUSY_avr GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/$tag -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-aux --graphs off > /dev/null 2> $out/$tag.err || echo "fail $tag"
done
