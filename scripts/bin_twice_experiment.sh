#!/bin/bash
# cost of one lattice evaluation inside the bin kernel: encode backward with the simplex computed once vs twice (-DPAG_DBG_BIN_TWICE)
cd "$GRAFT_REPO_ROOT"
python scripts/bench_encode.py 21 2>&1 | grep "bwd tables"
PAG_EXTRA_FLAGS="-DPAG_DBG_BIN_TWICE" python -m pagnerf_amd.build --force > /dev/null 2>&1
python scripts/bench_encode.py 21 2>&1 | grep "bwd tables"
