set -e
for i in 1 2; do
for mode in "--two-call" ""; do
  echo "== mode [$mode] run $i"
  python bench.py --rays 24576 --raymarch voxel --samples 2 --pose-opt --lin-assign $mode --channels all --no-aux --no-cpu-baseline --steps 40 --warmup 8 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"
  PAG_ASSIGN_SOLVER=scipy python bench.py --rays 24576 --raymarch voxel --samples 2 --pose-opt --lin-assign $mode --channels all --no-aux --no-cpu-baseline --steps 40 --warmup 8 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  host solver', d['ms_per_step'])"
done; done
python -m pytest tests/test_gpu_loss.py -x -q 2>&1 | tail -3
