cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03m
timeout 2700 python -m pytest tests -x -q -m gpu > gpurun_out/r03m/tests.log 2>&1; grep "passed\|failed" gpurun_out/r03m/tests.log | tail -3; grep -B30 "Error\|assert" gpurun_out/r03m/tests.log | grep "^E\|^tests/.*Error" | head -10
