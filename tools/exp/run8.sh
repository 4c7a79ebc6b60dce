cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03h
timeout 1800 python -m pytest tests/test_front_gpu.py tests/test_fullsize_gpu.py tests/test_pose_golden_gpu.py tests/test_stress_gpu.py -x -q -m gpu > gpurun_out/r03h/tests.log 2>&1; grep "passed\|failed" gpurun_out/r03h/tests.log
bash tools/exp/variants.sh r03h ""
for I in 3 5 6 8; do python bench.py --no-cpu --steps 12 --warmup 4 --inflight $I --h2d-steps 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('inflight $I: %.0f frames/s %.2f ms'%(d['value'], d['ms_per_step']))"; done | tee -a gpurun_out/r03h/variants.log
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03h/pipe -o p -- python bench.py --steps 10 --warmup 3 --no-cpu --h2d-steps 0 > /dev/null 2>&1
python - <<'PY' | tee -a gpurun_out/r03h/variants.log
import csv,glob
f=glob.glob("gpurun_out/r03h/pipe/**/*kernel_stats.csv", recursive=True)
print("pipelined kernel averages:")
for r in list(csv.DictReader(open(f[0])))[:10]:
    print("   %-52s %8.2f ms  calls %s"%(r["Name"][:52], float(r["AverageNs"])/1e6, r["Calls"]))
PY
find gpurun_out/r03h -name "*kernel_trace.csv" -delete
