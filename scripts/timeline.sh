#!/bin/bash
# Kernel timeline of one step on the GPU box:  bash scripts/timeline.sh <steps-back> [bench flags...]   -> gpurun_out/timeline.txt
back=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/tl_tmp; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --output-format csv -d $out -- python3 bench.py --steps 6 --warmup 4 --no-cpu-baseline --no-aux "$@" > $out/bench.json 2> $out/err
f=$(find $out -name "*kernel_trace.csv" | head -1)
python3 scripts/step_timeline.py $f adam_kernel $back > gpurun_out/timeline.txt
rm -rf $out
cat gpurun_out/timeline.txt
