#!/bin/bash
# GPU box: tools/exp/variants.sh <outdir> "<cflags A>" "<cflags B>" ...   -- rebuild with each flag set, time the default
# bench pipelined + serial, and keep the serial per-kernel table (rocprofv3 --kernel-trace --stats)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$1; shift; mkdir -p $OUT
i=0
for V in "$@"; do
  i=$((i+1))
  LF_EXTRA_CFLAGS="$V" python -m lineslam_amd.build --force > $OUT/build_$i.log 2>&1 || { echo "build failed: $V"; tail -5 $OUT/build_$i.log; continue; }
  echo "== variant $i: $V"
  timeout 600 python bench.py --no-cpu --no-config4 --no-legs --steps 10 --warmup 3 ${BENCH_EXTRA:-} 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pipelined %.0f frames/s %.2f ms'%(d['value'], d['ms_per_step']))"
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$i -o v -- python bench.py --steps 3 --warmup 1 --no-cpu --no-config4 --no-legs --inflight 1 ${BENCH_EXTRA:-} > $OUT/serial_$i.log 2>&1
  grep "^{" $OUT/serial_$i.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('serial %.0f frames/s %.2f ms'%(d['value'], d['ms_per_step']))"
  python - <<PY
import csv,glob
f=glob.glob("$OUT/stats_$i/**/*kernel_stats.csv", recursive=True)
if f:
    for r in list(csv.DictReader(open(f[0])))[:9]:
        print("   %-52s %8.2f ms"%(r["Name"][:52], float(r["AverageNs"])/1e6))
PY
  find $OUT/stats_$i -name "*kernel_trace.csv" -delete
done 2>&1 | tee $OUT/variants.log
