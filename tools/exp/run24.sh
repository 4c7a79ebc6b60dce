cd "$GRAFT_REPO_ROOT"
for cfg in "1147 4 12" "574 8 24" "383 12 36" "574 6 24" "287 16 48"; do
set -- $cfg
timeout 600 python bench.py --no-cpu --frames $1 --inflight $2 --steps $3 --warmup 4 --h2d-steps 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('frames $1 inflight $2: %.0f frames/s %.2f ms/step'%(d['value'], d['ms_per_step']))"
done
