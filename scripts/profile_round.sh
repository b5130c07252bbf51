#!/bin/bash
# Round profile set, run on the GPU box:  bash scripts/profile_round.sh <tag>   (outputs under gpurun_out/prof_<tag>/)
tag=${1:-rXX}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_$tag
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-aux > $out/bench_under_rocprof.json 2> $out/trace.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-aux --graphs off > /dev/null 2> $out/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-aux --graphs off > /dev/null 2> $out/write.err
python3 scripts/pmc_traffic.py $(find $out/fetch -name "*counter_collection.csv" | head -1) $(find $out/write -name "*counter_collection.csv" | head -1) $out/pmc_traffic_per_launch.json
cp $(find $out/trace -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
rm -rf $out/trace $out/fetch $out/write
python3 bench.py > $out/bench_default_run.json 2> $out/bench_default.err
tail -c 600 $out/bench_default_run.json
