cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03r
LF_EXTRA_CFLAGS="$1" python -m lineslam_amd.build --force > gpurun_out/r03r/build.log 2>&1 || tail -5 gpurun_out/r03r/build.log
timeout 900 python -m pytest tests/test_pair_gpu.py tests/test_pose_golden_gpu.py -x -q 2>&1 | tail -4
timeout 600 python bench.py --no-cpu --steps 10 --warmup 3 --h2d-steps 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pipelined %.0f frames/s %.2f ms'%(d['value'], d['ms_per_step']), d['serial']['stage_ms'])"
