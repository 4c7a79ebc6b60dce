cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03g
timeout 2400 python -m pytest tests/test_loopclosure_gpu.py tests/test_hybrid_gpu.py tests/test_exchange_gpu.py -x -q -m gpu > gpurun_out/r03g/tests.log 2>&1; tail -5 gpurun_out/r03g/tests.log
( time python bench.py --steps 10 --warmup 3 --no-cpu ) > gpurun_out/r03g/bench_default.log 2>&1; grep "^{" gpurun_out/r03g/bench_default.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default', d['value'], d['ms_per_step'], 'h2d', d['value_including_h2d'], d['quality'])"; grep real gpurun_out/r03g/bench_default.log
python tools/bench_config4.py > gpurun_out/r03g/config4_match.log 2>&1; grep "^{" gpurun_out/r03g/config4_match.log | tail -1 | cut -c1-400
python tools/bench_config4.py --pose > gpurun_out/r03g/config4_pose.log 2>&1; grep "^{" gpurun_out/r03g/config4_pose.log | tail -1 | cut -c1-400
nproc
