#!/bin/bash
# A/B of an environment switch on ONE GPU box:  bash scripts/env_ab.sh "<kernel-grep>" VAR valueA valueB ...   (BENCH_ARGS = extra bench flags)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
pat=$1; var=$2; shift 2
for round in 1 2; do
for v in "$@"; do
  export $var=$v
  out=gpurun_out/ab_tmp; rm -rf $out; mkdir -p $out
  rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-aux $BENCH_ARGS > $out/bench.json 2> $out/err
  f=$(find $out -name "*kernel_stats.csv" | head -1)
  echo "[$var=$v] $(python3 - $f "$pat" <<'PY'
import csv,sys,re
rows=list(csv.DictReader(open(sys.argv[1])))
print("  ".join("%s %.1f" % (re.sub(r".*::|<.*|\(.*","",r["Name"])[:22], float(r["AverageNs"])/1e3) for r in rows if re.search(sys.argv[2], r["Name"])))
PY
) step(under rocprof) $(python3 -c "import json;print(json.loads(open('$out/bench.json').read().strip().splitlines()[-1])['ms_per_step'])")"
  rm -rf $out
  echo "   plain: $(python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-aux $BENCH_ARGS 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])")"
done; done
