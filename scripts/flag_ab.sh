#!/bin/bash
# A/B of build flags on ONE GPU box:  bash scripts/flag_ab.sh "<flagsA>" "<flagsB>" [kernel-grep]   (alternates A B A B; prints the kernels' average us)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
pat=${3:-reduce_kernel|bin_kernel}
for round in 1 2; do
  for fl in "$1" "$2"; do
    export PAG_EXTRA_FLAGS="$fl"
    out=gpurun_out/ab_tmp; rm -rf $out; mkdir -p $out
    rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-aux $BENCH_ARGS > $out/bench.json 2> $out/err
    f=$(find $out -name "*kernel_stats.csv" | head -1)
    echo "[$fl] $(python3 - $f "$pat" <<'PY'
import csv,sys,re
rows=list(csv.DictReader(open(sys.argv[1])))
print("  ".join("%s %.1f" % (re.sub(r".*::|<.*|\(.*","",r["Name"])[:18], float(r["AverageNs"])/1e3) for r in rows if re.search(sys.argv[2], r["Name"])))
PY
) step $(python3 -c "import json;print(json.loads(open('$out/bench.json').read().strip().splitlines()[-1])['ms_per_step'])")"
    rm -rf $out
  done
done
