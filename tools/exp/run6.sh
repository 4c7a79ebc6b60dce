cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03f
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r03f/tests.log 2>&1; tail -5 gpurun_out/r03f/tests.log
bash tools/exp/variants.sh r03f ""
