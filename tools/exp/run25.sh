cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03u
timeout 1200 python -m pytest tests/test_hybrid_gpu.py tests/test_pose_golden_gpu.py tests/test_node_gpu.py tests/test_operators_gpu.py tests/test_orb_gpu.py tests/test_pair_gpu.py -x -q 2>&1 > gpurun_out/r03u/t.log; tail -5 gpurun_out/r03u/t.log | cut -c1-300
