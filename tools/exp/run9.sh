cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03i
timeout 1800 python -m pytest tests/test_front_gpu.py tests/test_fullsize_gpu.py tests/test_pose_golden_gpu.py -x -q -m gpu > gpurun_out/r03i/tests.log 2>&1; grep "passed\|failed" gpurun_out/r03i/tests.log
BENCH_EXTRA="--h2d-steps 0" bash tools/exp/variants.sh r03i "-DLF_MLE_WAVES=3" "-DLF_MLE_WAVES=4"
