"""python tools/sq_summary.py <dir> <tag>: per-kernel sums of every counter found in the rocprofv3 counter_collection
CSVs under <dir>/pass*/, averaged per dispatch, plus the derived ratios quoted in DESIGN.md:
  valu_busy   = SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES-per-SIMD ...   (see the columns written)"""
import csv, glob, json, os, sys
from collections import defaultdict

d, tag = sys.argv[1], sys.argv[2]
tot = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
for fn in glob.glob(os.path.join(d, "pass*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"].split("(")[0]
        c = r["Counter_Name"]
        tot[k][c] += float(r["Counter_Value"])
        cnt[k][c] += 1
names = sorted({c for k in tot for c in tot[k]})
rows = sorted(tot, key=lambda k: -tot[k].get("SQ_WAVE_CYCLES", 0))
with open(os.path.join(d, tag + "_sq_counters_by_kernel.csv"), "w") as f:
    f.write("Kernel,Dispatches," + ",".join(names) + "\n")
    for k in rows:
        n = max(max(cnt[k].values()), 1)
        f.write("%s,%d," % (k, n) + ",".join("%.0f" % (tot[k].get(c, 0.0) / max(cnt[k].get(c, 1), 1)) for c in names) + "\n")
der = {}
for k in rows:
    t = {c: tot[k][c] / max(cnt[k][c], 1) for c in tot[k]}
    wc = t.get("SQ_WAVE_CYCLES", 0)
    if not wc:
        continue
    g = lambda c: t.get(c, 0.0)
    der[k] = {
        "wave_cycles": wc, "waves": g("SQ_WAVES"),
        "frac_active_any": g("SQ_ACTIVE_INST_ANY") / wc, "frac_active_valu": g("SQ_ACTIVE_INST_VALU") / wc,
        "frac_wait_any": g("SQ_WAIT_ANY") / wc, "frac_wait_inst_any": g("SQ_WAIT_INST_ANY") / wc,
        "frac_active_lds": g("SQ_ACTIVE_INST_LDS") / wc, "frac_active_vmem": g("SQ_ACTIVE_INST_VMEM") / wc,
        "frac_active_scalar": g("SQ_ACTIVE_INST_SCA") / wc,
        "insts_valu": g("SQ_INSTS_VALU"), "insts_salu": g("SQ_INSTS_SALU"), "insts_vmem_rd": g("SQ_INSTS_VMEM_RD"),
        "insts_vmem_wr": g("SQ_INSTS_VMEM_WR"), "insts_lds": g("SQ_INSTS_LDS"),
        # chip-level: VALU instruction issue slots used / available.  One SIMD issues at most one VALU instruction
        # per cycle (an fp64 op occupies the pipe for more than one); available = busy cycles x 4 SIMDs x CUs
        "busy_cycles": g("SQ_BUSY_CYCLES"), "gui_active": g("GRBM_GUI_ACTIVE"),
        "l2_hit": (g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum"))) if g("TCC_HIT_sum") + g("TCC_MISS_sum") else None,
    }
json.dump(der, open(os.path.join(d, tag + "_sq_derived.json"), "w"), indent=1)
for k in rows[:8]:
    if k in der:
        print(k, json.dumps({a: (round(b, 4) if isinstance(b, float) and b < 10 else b) for a, b in der[k].items()}))
