#!/bin/bash
# Run on the GPU box (through gpurun): FETCH_SIZE / WRITE_SIZE (separate passes) of the 1146-pair launch of k_match / k_pose.
#   tools/pair_pmc.sh <tag>      -> gpurun_out/pairpmc_<tag>/<tag>_pmc_fetch_write_by_kernel.csv
set -u
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pairpmc_$TAG
mkdir -p $OUT
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$C -o p -- python tools/pair_once.py 2 > $OUT/pmc_$C.log 2>&1
done
python tools/pmc_summary.py $OUT $TAG | grep "Kernel\|k_pose\|k_match"
find $OUT -name "*kernel_trace.csv" -delete
