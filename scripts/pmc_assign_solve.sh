#!/bin/bash
# SQ counters of assign_solve_kernel on scripts/bench_assign_solve.py's cases: bash scripts/pmc_assign_solve.sh <outdir>; then scripts/pmc_summary.py <outdir> assign_solve
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=${1:-gpurun_out/pmc_solve}
mkdir -p $out
for set in "VALUBusy SALUBusy" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_SALU SQ_WAIT_ANY"; do
  tag=$(echo $set | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/$tag -- python3 scripts/bench_assign_solve.py > /dev/null 2> $out/$tag.err || echo "fail $tag"
done
python3 scripts/pmc_summary.py $out assign_solve > $out/summary.txt
cat $out/summary.txt
find $out -name "*.csv" -delete
