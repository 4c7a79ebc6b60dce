#!/bin/bash
# GPU box: tools/exp/ab.sh <outdir> <rounds> libA.so libB.so ...   -- same-box A/B of prebuilt library variants (LF_LIB): the default
# bench, pipelined, alternating the variants <rounds> times (box-to-box spread is +-3 %, so variants are only compared on one box)
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$1; shift; R=$1; shift; mkdir -p $OUT
for r in $(seq 1 $R); do
  for L in "$@"; do
    LF_LIB=$L timeout 600 python bench.py --no-cpu --no-config4 --no-legs --steps ${STEPS:-10} --warmup 3 ${BENCH_EXTRA:-} 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d.get('serial') or {}; print('$L  round $r  pipelined %.0f frames/s %.2f ms   serial %s'%(d['value'], d['ms_per_step'], {k:round(v,1) for k,v in (s.get('stage_ms') or {}).items()}))"
  done
done 2>&1 | tee $OUT/ab.log
