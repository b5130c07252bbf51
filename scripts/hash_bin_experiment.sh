cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for fl in "" "-DPAG_BIN_WAVES_HASH=8"; do
  PAG_EXTRA_FLAGS="$fl" python -m pagnerf_amd.build --force > /dev/null 2>&1
  echo "FLAGS [$fl]"
  PAG_EXTRA_FLAGS="$fl" python bench.py --grid hash --steps 30 --warmup 5 --no-cpu-baseline --no-aux 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print(d['ms_per_step'], d['value'], k['hash_encode_bwd_set']['ms_per_step'])"
done
done
python -m pagnerf_amd.build --force > /dev/null 2>&1
python -m pytest tests -m gpu -q -x 2>&1 | tail -1
