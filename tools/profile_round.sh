#!/bin/bash
# Run on the GPU box (through gpurun): kernel-trace stats of the default bench command, then two separate PMC
# passes (FETCH_SIZE, WRITE_SIZE cannot share a pass: TCC slots) of ONE serial pass of the same workload.
#   tools/profile_round.sh <tag>      -> gpurun_out/prof_<tag>/...
set -u
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o $TAG -- python bench.py --no-cpu --no-config4 --no-legs --h2d-steps 0 > $OUT/bench_stats.log 2>&1
grep "^{\"metric\"" $OUT/bench_stats.log | tail -1 > $OUT/bench_line_during_profile.json
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_serial -o ${TAG}s -- python bench.py --steps 3 --warmup 1 --no-cpu --no-config4 --no-legs --inflight 1 --h2d-steps 0 > $OUT/bench_stats_serial.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$C -o p -- python bench.py --steps 1 --warmup 0 --no-cpu --no-config4 --no-legs --inflight 1 --h2d-steps 0 > $OUT/bench_pmc_$C.log 2>&1
done
python tools/pmc_summary.py $OUT $TAG
find $OUT -name "*kernel_trace.csv" -delete     # large; the stats tables are what is kept
ls -la $OUT $OUT/*
