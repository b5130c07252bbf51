cd $GRAFT_REPO_ROOT
for j in 0 1 2; do
  PAG_EXTRA_FLAGS=-DPAG_DBG_ONLY_J=$j python -m pagnerf_amd.build --force > /dev/null 2>&1
  echo "ONLY_J=$j"; PAG_EXTRA_FLAGS=-DPAG_DBG_ONLY_J=$j python scripts/bench_encode.py 2>/dev/null | head -1
done
python -m pagnerf_amd.build --force > /dev/null 2>&1
echo full; python scripts/bench_encode.py 2>/dev/null | head -1
