cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03y
timeout 2700 python -m pytest tests -x -q -m gpu > gpurun_out/r03y/tests.log 2>&1; grep "passed\|failed" gpurun_out/r03y/tests.log | tail -3; grep -B30 "Error\|assert" gpurun_out/r03y/tests.log | grep "^E\|^tests/.*Error" | head -10
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/r03y/bench.log 2>&1; grep "^{" gpurun_out/r03y/bench.log | tail -1
