#!/bin/bash
# A/B of library variants (scripts/build_variant.sh) on ONE GPU box:  bash scripts/variant_ab.sh "<kernel-grep>" <tagA> <tagB> ...   ("-" = the regular library)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
pat=$1; shift
for round in 1 2; do
for v in "$@"; do
  if [ "$v" == "-" ]; then unset PAG_LIB_VARIANT; else export PAG_LIB_VARIANT=$v; fi
  out=gpurun_out/ab_tmp; rm -rf $out; mkdir -p $out
  rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-aux $BENCH_ARGS > $out/bench.json 2> $out/err
  f=$(find $out -name "*kernel_stats.csv" | head -1)
  echo "[$v] $(python3 - $f "$pat" <<'PY'
import csv,sys,re
rows=list(csv.DictReader(open(sys.argv[1])))
print("  ".join("%s %.1f" % (re.sub(r".*::|<.*|\(.*","",r["Name"])[:18], float(r["AverageNs"])/1e3) for r in rows if re.search(sys.argv[2], r["Name"])))
PY
) step $(python3 -c "import json;print(json.loads(open('$out/bench.json').read().strip().splitlines()[-1])['ms_per_step'])")"
  rm -rf $out
done; done
