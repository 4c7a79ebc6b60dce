cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03u
LF_EXTRA_CFLAGS="-DLF_POSE_PROFILE=1 $1" python -m lineslam_amd.build --force > gpurun_out/r03u/build.log 2>&1 || tail -5 gpurun_out/r03u/build.log
timeout 900 python bench.py --no-cpu --steps 1 --warmup 0 --h2d-steps 0 --points --frames 64 --inflight 1 2>&1 | grep "k_pose_hybrid prof" | tail -2
