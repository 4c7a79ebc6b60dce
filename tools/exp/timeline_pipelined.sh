#!/bin/bash
# GPU box: tools/exp/timeline_pipelined.sh <tag> [bench args]: kernel trace of the default (pipelined) bench + text Gantt per stream
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$1; shift; mkdir -p $OUT
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python bench.py --no-cpu --no-config4 --no-legs --steps 8 --warmup 2 --h2d-steps 0 "$@" > $OUT/bench.log 2>&1
F=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
for off in 0 500 1000 1500 2000 2500; do echo "== window ending $off ms before the end"; python tools/timeline.py $F 500 $off | grep -v "^columns" | head -12; done > $OUT/timeline.txt
tail -1 $OUT/bench.log | cut -c1-200
cat $OUT/timeline.txt | cut -c1-200
rm -rf $OUT/trace
