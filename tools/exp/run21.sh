cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03r
for V in "$@"; do
echo "== $V"
LF_EXTRA_CFLAGS="$V" python -m lineslam_amd.build --force > gpurun_out/r03r/build.log 2>&1 || tail -5 gpurun_out/r03r/build.log
for I in ${INFL:-4}; do
timeout 600 python bench.py --no-cpu --steps 12 --warmup 3 --h2d-steps 0 --inflight $I 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('inflight $I: pipelined %.0f frames/s %.2f ms'%(d['value'], d['ms_per_step']))"
done
done
