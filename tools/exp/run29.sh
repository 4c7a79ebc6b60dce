cd "$GRAFT_REPO_ROOT"
bash tools/profile_round.sh r03c > gpurun_out/prof_r03c.log 2>&1
bash tools/profile_sq.sh r03c > gpurun_out/sq_r03c.log 2>&1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03c_points gpurun_out/r03c_edlines
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03c_points/stats -o v -- python bench.py --points --steps 3 --warmup 1 --no-cpu --inflight 1 --h2d-steps 0 > gpurun_out/r03c_points/serial.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03c_edlines/stats -o v -- python bench.py --detector edlines --steps 3 --warmup 1 --no-cpu --inflight 1 --h2d-steps 0 > gpurun_out/r03c_edlines/serial.log 2>&1
find gpurun_out/r03c_points gpurun_out/r03c_edlines -name "*kernel_trace.csv" -delete
timeout 900 python bench.py --points --no-cpu --h2d-steps 0 2>/dev/null | tail -1 > gpurun_out/r03c_points/bench_line.json
timeout 900 python bench.py --detector edlines --no-cpu --h2d-steps 0 2>/dev/null | tail -1 > gpurun_out/r03c_edlines/bench_line.json
timeout 600 python tools/bench_config4.py > gpurun_out/r03c_config4.json 2>/dev/null
timeout 600 python tools/bench_config4.py --pose > gpurun_out/r03c_config4_pose.json 2>/dev/null
timeout 1200 python bench.py > gpurun_out/r03c_bench_default.log 2>&1; grep "^{" gpurun_out/r03c_bench_default.log | tail -1 | cut -c1-200
ls gpurun_out/prof_r03c gpurun_out/sq_r03c | head -40
