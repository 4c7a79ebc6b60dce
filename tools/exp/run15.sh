cd "$GRAFT_REPO_ROOT"
bash tools/profile_round.sh r03b > gpurun_out/prof_r03b.log 2>&1; tail -30 gpurun_out/prof_r03b.log | head -40
bash tools/profile_sq.sh r03b > gpurun_out/sq_r03b.log 2>&1; tail -25 gpurun_out/sq_r03b.log
python bench.py > gpurun_out/bench_default_r03b.log 2>gpurun_out/bench_default_r03b.err; tail -1 gpurun_out/bench_default_r03b.log | cut -c1-300
find gpurun_out/prof_r03b gpurun_out/sq_r03b -name "*counter_collection.csv" -size +10M -delete
