#!/bin/bash
# A/B of compile-time kernel variants on one box:  bash scripts/flag_experiment.sh "<flags A>" "<flags B>" ...   ("" = default build)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for fl in "$@"; do
  PAG_EXTRA_FLAGS="$fl" python -m pagnerf_amd.build --force > /dev/null 2>&1
  echo "FLAGS [$fl]"
  PAG_EXTRA_FLAGS="$fl" python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-aux 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['kernel_ms_per_step']['permuto_encode_bwd_set']['ms_per_step'])"
done
done
python -m pagnerf_amd.build --force > /dev/null 2>&1
