#!/bin/bash
# Quick per-kernel times on the GPU box:  bash scripts/quick_stats.sh <tag> [bench flags...]  -> gpurun_out/qs_<tag>.csv (top of kernel_stats)
tag=${1:-x}; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/qs_$tag
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-aux "$@" > $out/bench.json 2> $out/trace.err
cp $(find $out/trace -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
rm -rf $out/trace
python3 - $out/kernel_stats.csv <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:22]:
    print("%-60s calls %5s avg_us %9.1f pct %5s" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
