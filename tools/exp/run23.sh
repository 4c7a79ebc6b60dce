cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03r
for V in "$@"; do
echo "== $V"
LF_EXTRA_CFLAGS="$V" python -m lineslam_amd.build --force > gpurun_out/r03r/build.log 2>&1 || tail -5 gpurun_out/r03r/build.log
timeout 900 python bench.py --no-cpu --steps 8 --warmup 2 --h2d-steps 0 --points 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('points: pipelined %.0f frames/s %.2f ms'%(d['value'], d['ms_per_step']), 'serial %.1f'%d['serial']['ms_per_step'], d['quality']['pairs_over_a_capacity'], d.get('points',{}).get('point_matches_per_pair'))"
done
