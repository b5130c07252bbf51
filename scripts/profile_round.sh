#!/bin/bash
# Round profile set, run on the GPU box:  bash scripts/profile_round.sh <tag> <commit>   (outputs under gpurun_out/prof_<tag>/; <commit> = `git rev-parse --short HEAD` of
# the tree that was sent - the box has no .git - stamped into the PMC file as `_commit`)
tag=${1:-rXX}
export PAG_COMMIT=${2:-unknown}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_$tag
mkdir -p $out
# headline (configs[1]): kernel stats under graph replay, PMC traffic passes on the eager path
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-aux > $out/bench_under_rocprof.json 2> $out/trace.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-aux --graphs off > /dev/null 2> $out/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-aux --graphs off > /dev/null 2> $out/write.err
python3 scripts/pmc_traffic.py $(find $out/fetch -name "*counter_collection.csv" | head -1) $(find $out/write -name "*counter_collection.csv" | head -1) $out/pmc_traffic_per_launch.json
cp $(find $out/trace -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
rm -rf $out/trace $out/fetch $out/write
# the step best.yaml runs (6 images x 4096 rays, pose optimisation): kernel stats of its three regimes
for reg in "dense_rgbd:--channels rgbd" "post_prune_rgbd:--raymarch voxel --channels rgbd" "post_prune_all_assign:--raymarch voxel --channels all --lin-assign --two-call" "post_prune_all_assign_one_backward:--raymarch voxel --channels all --lin-assign"; do
  name=${reg%%:*}; flags=${reg#*:}
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/t_$name -- python3 bench.py --pose-opt --rays 24576 $flags --steps 5 --warmup 3 --no-cpu-baseline --no-aux > $out/bench_${name}_under_rocprof.json 2> $out/t_$name.err
  cp $(find $out/t_$name -name "*kernel_stats.csv" | head -1) $out/kernel_stats_best_yaml_$name.csv
  rm -rf $out/t_$name
done
python3 bench.py > $out/bench_default_run.json 2> $out/bench_default.err
cp bench_detail.json $out/bench_detail.json        # the full record behind the compact line (round 6: the line itself stays under 4 KB)
tail -c 600 $out/bench_default_run.json
