#!/bin/bash
# GPU box: how long do the host->device copies of the h2d leg take inside the pipelined run? (memory-copy trace)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$1; mkdir -p $OUT
timeout 900 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/trace -o t -- python bench.py --no-cpu --no-config4 --no-legs --steps 2 --warmup 1 --h2d-steps 6 > $OUT/bench.log 2>&1
F=$(find $OUT/trace -name "*memory_copy_trace.csv" | head -1)
python - <<PY
import csv
rows=list(csv.DictReader(open("$F")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
t0=int(rows[0]["Start_Timestamp"])
long=[r for r in rows if int(r["End_Timestamp"])-int(r["Start_Timestamp"]) > 2e6]
print(len(rows), "copies,", len(long), "longer than 2 ms")
for r in long[-16:]:
    print(r["Direction"], "stream", r["Stream_Id"], "start %.1f ms  duration %.1f ms" % ((int(r["Start_Timestamp"])-t0)/1e6, (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6))
PY
import csv
rows=list(csv.DictReader(open("$F")))
print(rows[0].keys())
big=[r for r in rows if int(r.get("Bytes", r.get("Size", 0)) or 0) > 100e6]
print(len(rows), "copies,", len(big), "large")
for r in big[-12:]:
    b=int(r.get("Bytes", r.get("Size", 0))); d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6
    print(r.get("Direction"), "%.0f MB %.1f ms %.1f GB/s start %.1f" % (b/1e6, d, b/d/1e6, int(r["Start_Timestamp"])/1e6 % 100000))
PY
grep "^{" $OUT/bench.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['value_including_h2d'], d['including_h2d']['ms_per_step'])"
rm -rf $OUT/trace
