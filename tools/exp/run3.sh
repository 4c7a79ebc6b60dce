cd "$GRAFT_REPO_ROOT"
bash tools/exp/variants.sh r03c "-DLF_MLE_WAVES=3 -DLF_SWEEP_PRIO=0 -DLF_POSE_PRIO=0" "-DLF_MLE_WAVES=3 -DLF_SWEEP_PRIO=3 -DLF_POSE_PRIO=0" "-DLF_MLE_WAVES=3 -DLF_SWEEP_PRIO=3 -DLF_POSE_PRIO=2" "-DLF_MLE_WAVES=2 -DLF_SWEEP_PRIO=3 -DLF_POSE_PRIO=2"
# pipelined timeline of the last variant
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r03c/trace -o t -- python bench.py --steps 8 --warmup 2 --no-cpu > /dev/null 2>&1
python tools/timeline.py $(find gpurun_out/r03c/trace -name "*kernel_trace.csv" | head -1) 450 > gpurun_out/r03c/timeline_pipelined.txt 2>&1
find gpurun_out/r03c/trace -name "*kernel_trace.csv" -delete
