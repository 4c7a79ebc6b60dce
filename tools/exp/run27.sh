cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03u
timeout 1200 python -m pytest tests/test_hybrid_gpu.py tests/test_pose_golden_gpu.py tests/test_node_gpu.py tests/test_pair_gpu.py tests/test_operators_gpu.py tests/test_orb_gpu.py tests/test_lsd_gpu.py -x -q 2>&1 > gpurun_out/r03u/t.log; tail -3 gpurun_out/r03u/t.log | cut -c1-300
timeout 900 python bench.py --no-cpu --steps 8 --warmup 2 --h2d-steps 0 --points 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('points: pipelined %.0f frames/s %.2f ms'%(d['value'], d['ms_per_step']), 'serial', d['serial']['stage_ms']['match_pose'], d['quality']['ate_rmse_m_vs_ground_truth'])"
