cd "$GRAFT_REPO_ROOT"
BENCH_EXTRA="" bash tools/exp/variants.sh r03d "-DLF_SWEEP_WAVES=4 -DLF_MLE_WAVES=3" "-DLF_SWEEP_WAVES=4 -DLF_MLE_WAVES=4" "-DLF_SWEEP_WAVES=5 -DLF_MLE_WAVES=3"
for I in 2 6 8; do python bench.py --no-cpu --steps 10 --warmup 3 --inflight $I 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('inflight $I (last variant): %.0f frames/s %.2f ms'%(d['value'], d['ms_per_step']))"; done | tee -a gpurun_out/r03d/variants.log
