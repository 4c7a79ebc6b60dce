cd "$GRAFT_REPO_ROOT"
BENCH_EXTRA="--points --h2d-steps 0" bash tools/exp/variants.sh r03l_points ""
BENCH_EXTRA="--detector edlines --h2d-steps 0" bash tools/exp/variants.sh r03l_edlines ""
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d gpurun_out/r03l_points/pmc_$C -o p -- python bench.py --points --steps 1 --warmup 0 --no-cpu --inflight 1 --h2d-steps 0 > /dev/null 2>&1
done
python tools/pmc_summary.py gpurun_out/r03l_points r03l_points 2>&1 | tail -25
find gpurun_out/r03l_points -name "*kernel_trace.csv" -delete; find gpurun_out/r03l_points -name "*counter_collection.csv" -size +20M -delete
