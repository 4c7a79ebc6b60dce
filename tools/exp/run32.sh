cd "$GRAFT_REPO_ROOT"
for V in "" "-DLF_DEFER_IMPROVE=0"; do
echo "== $V"
LF_EXTRA_CFLAGS="$V" python -m lineslam_amd.build --force > /dev/null 2>&1
for I in 3 4 5 6 8; do
timeout 600 python bench.py --no-cpu --steps 24 --warmup 8 --h2d-steps 0 --inflight $I 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('inflight $I: %.0f frames/s %.2f ms'%(d['value'], d['ms_per_step']))"
done
done
