cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03z
LF_BENCH_FORCE_EXCHANGE=1 timeout 900 python bench.py --no-cpu --steps 6 --h2d-steps 0 --scaling strong > gpurun_out/r03z/strong.log 2>&1; tail -5 gpurun_out/r03z/strong.log | cut -c1-600
LF_BENCH_FORCE_EXCHANGE=1 timeout 900 python bench.py --no-cpu --steps 6 --h2d-steps 0 > gpurun_out/r03z/weak.log 2>&1; tail -3 gpurun_out/r03z/weak.log | cut -c1-300
