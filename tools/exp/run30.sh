cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03x
timeout 1500 python -m pytest tests/test_lsd_gpu.py tests/test_edges_gpu.py tests/test_front_gpu.py tests/test_bigframe_gpu.py tests/test_fullsize_gpu.py -x -q 2>&1 > gpurun_out/r03x/t.log; tail -4 gpurun_out/r03x/t.log | cut -c1-300
timeout 600 python bench.py --no-cpu --steps 10 --warmup 3 --h2d-steps 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pipelined %.0f frames/s %.2f ms'%(d['value'], d['ms_per_step']), d['serial']['stage_ms'], d['quality']['ate_rmse_m_vs_ground_truth'])"
