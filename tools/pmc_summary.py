"""python tools/pmc_summary.py <dir> <tag>: per-kernel FETCH_SIZE / WRITE_SIZE (KB per dispatch, averaged over the dispatches of the run)
from the two rocprofv3 counter_collection CSVs under <dir>/pmc_*, plus the sweep's HBM bytes per launch as JSON."""
import csv, glob, json, os, sys
from collections import defaultdict

d, tag = sys.argv[1], sys.argv[2]
tot = {"FETCH_SIZE": defaultdict(float), "WRITE_SIZE": defaultdict(float)}
cnt = defaultdict(int)
for c in tot:
    for fn in glob.glob(os.path.join(d, "pmc_" + c, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(fn)):
            if r.get("Counter_Name") != c:
                continue
            k = r["Kernel_Name"].split("(")[0]
            tot[c][k] += float(r["Counter_Value"])
            if c == "FETCH_SIZE":
                cnt[k] += 1
rows = sorted(tot["FETCH_SIZE"], key=lambda k: -(tot["FETCH_SIZE"][k] + tot["WRITE_SIZE"].get(k, 0)))
with open(os.path.join(d, tag + "_pmc_fetch_write_by_kernel.csv"), "w") as f:
    f.write("Kernel,Dispatches,FETCH_SIZE_KB_per_dispatch,WRITE_SIZE_KB_per_dispatch\n")
    for k in rows:
        f.write("%s,%d,%.1f,%.1f\n" % (k, cnt[k], tot["FETCH_SIZE"][k] / max(cnt[k], 1), tot["WRITE_SIZE"].get(k, 0.0) / max(cnt[k], 1)))
sw = [k for k in rows if k.startswith("k_lsd_sweep")]
if sw:
    k = sw[0]
    n = max(cnt[k], 1)
    import sys as _s, os as _o
    _s.path.insert(0, _o.path.dirname(_o.path.dirname(_o.path.abspath(__file__))))
    from bench import csrc_sha256
    js = {"csrc_sha256": csrc_sha256(), "frames": 1147, "kernel": k, "fetch_kb": tot["FETCH_SIZE"][k] / n, "write_kb": tot["WRITE_SIZE"].get(k, 0.0) / n,
          "hbm_bytes_per_launch": (tot["FETCH_SIZE"][k] + tot["WRITE_SIZE"].get(k, 0.0)) / n * 1024.0,
          "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of one serial bench pass (--inflight 1), "
                  "(FETCH+WRITE)*1024 per launch; the sweep's accesses are narrow 16-byte / 1-byte gathers, so the gfx950 x2 correction "
                  "for wide coalesced reads (MI355X_MICROARCH.md, HBM) is NOT applied (uncalibrated for this pattern)"}
    json.dump(js, open(os.path.join(d, tag + "_sweep_pmc.json"), "w"), indent=1)
print(open(os.path.join(d, tag + "_pmc_fetch_write_by_kernel.csv")).read())
