#!/bin/bash
# GPU box: tools/config4_pmc.sh <tag> -- FETCH_SIZE / WRITE_SIZE (separate passes) of k_match in the BASELINE config 4 shape (1 query vs
# 256 key frames, one 256-pair launch) -> gpurun_out/c4pmc_<tag>/<tag>_config4_match_pmc.json (bench.py: config4.matching_only.roofline.traffic)
set -u
TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/c4pmc_$TAG
mkdir -p $OUT
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$C -o p -- python tools/config4_once.py 4 > $OUT/pmc_$C.log 2>&1
done
python - <<PY
import csv, glob, json, os, sys
sys.path.insert(0, ".")
from bench import csrc_sha256
tot = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    v = []
    for fn in glob.glob("$OUT/pmc_" + c + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            if r.get("Counter_Name") == c and r["Kernel_Name"].startswith("void k_match<false>"):
                v.append(float(r["Counter_Value"]))
    tot[c] = sum(v) / max(len(v), 1)
    print(c, len(v), "launches", tot[c], "KB per launch")
json.dump({"csrc_sha256": csrc_sha256(), "kernel": "k_match<false>", "pairs_per_launch": 256, "fetch_kb": tot["FETCH_SIZE"], "write_kb": tot["WRITE_SIZE"],
           "hbm_bytes_per_launch": (tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024.0,
           "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of tools/config4_once.py (4 launches each), (FETCH + WRITE) * 1024 per launch"},
          open("$OUT/${TAG}_config4_match_pmc.json", "w"), indent=1)
PY
find $OUT -name "*kernel_trace.csv" -delete
cat $OUT/${TAG}_config4_match_pmc.json
