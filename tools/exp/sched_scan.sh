#!/bin/bash
# GPU box: tools/exp/sched_scan.sh <outdir> "<pose_cus list>" "<stagger list>" "<inflight list>"  -- default bench under the scheduling controls
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$1; mkdir -p $OUT
for nfl in $4; do for p in $2; do for d in $3; do
  timeout 300 python bench.py --no-cpu --no-config4 --steps ${STEPS:-10} --warmup 4 --inflight $nfl --h2d-steps 0 --pose-cus $p --stagger $d ${BENCH_EXTRA:-} 2>$OUT/err.txt | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pose_cus %3d stagger %d inflight %d  %.0f frames/s %.2f ms  sweep %.1f front %.1f pair %.1f' % ($p, $d, $nfl, d['value'], d['ms_per_step'], d['stage_ms']['lsd_sweep'], d['stage_ms']['lines3d_msld_mle'], d['stage_ms']['match_pose']))" || tail -3 $OUT/err.txt
done; done; done 2>&1 | tee $OUT/scan.log
