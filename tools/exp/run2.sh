cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03b
timeout 900 python -m pytest tests/test_front_gpu.py tests/test_relmotion_gpu.py tests/test_pose_golden_gpu.py tests/test_operators_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu > gpurun_out/r03b/tests.log 2>&1; tail -5 gpurun_out/r03b/tests.log
bash tools/exp/variants.sh r03b "-DLF_MLE_WAVES=2 -DLF_MLE_PAIR64=1" "-DLF_MLE_WAVES=2 -DLF_MLE_PAIR64=0" "-DLF_MLE_WAVES=3 -DLF_MLE_PAIR64=1"
