#!/bin/bash
# GPU box: tools/exp/grid_sweep.sh <outdir> "<grids>" "<inflights>"   -- LF_SWEEP_GRID x passes in flight, default bench, one box
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$1; mkdir -p $OUT
for nfl in $3; do
  for g in $2; do
    LF_SWEEP_GRID=$g timeout 600 python bench.py --no-cpu --no-config4 --steps ${STEPS:-10} --warmup 3 --inflight $nfl --h2d-steps 0 ${BENCH_EXTRA:-} 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('grid %5s inflight %d  %.0f frames/s %.2f ms  sweep %.1f front %.1f pair %.1f' % ('$g', $nfl, d['value'], d['ms_per_step'], d['stage_ms']['lsd_sweep'], d['stage_ms']['lines3d_msld_mle'], d['stage_ms']['match_pose']))"
  done
done 2>&1 | tee $OUT/grid.log
