cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
for V in "$@"; do
LF_EXTRA_CFLAGS="$V" python -m lineslam_amd.build --force > /dev/null 2>&1
timeout 600 python bench.py --no-cpu --steps 24 --warmup 8 --h2d-steps 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$V] %.0f frames/s %.2f ms'%(d['value'], d['ms_per_step']))"
done
done
