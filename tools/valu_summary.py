"""python tools/valu_summary.py <sq_counters_by_kernel.csv> <out.json> [note]: per-kernel VALU utilisation of the chip and
the wait / issue fractions of the wave cycles, from the SQ / GRBM counter table of tools/profile_sq.sh (read by bench.py
for roofline.valu)."""
import csv, json, sys
src, out = sys.argv[1], sys.argv[2]
note = sys.argv[3] if len(sys.argv) > 3 else ""
ker = {}
lines = open(src).read().strip().split("\n")
hdr = lines[0].split(",")
for ln in lines[1:]:
    r = dict(zip(hdr, ln.rsplit(",", len(hdr) - 1)))      # (kernel names contain commas: split from the right)
    k = r["Kernel"].replace("void ", "")
    g = lambda n: float(r.get(n) or 0)
    if g("GRBM_GUI_ACTIVE") <= 0 or g("SQ_WAVE_CYCLES") <= 0 or k in ker:
        continue
    hit, miss = g("TCC_HIT_sum"), g("TCC_MISS_sum")
    ker[k] = {"valu_busy_chip": g("SQ_ACTIVE_INST_VALU") * 4 / (g("GRBM_GUI_ACTIVE") / 8 * 1024),
              "wave_wait_frac": g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"),
              "wave_issue_stall_frac": g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES"),
              "wave_valu_frac": g("SQ_ACTIVE_INST_VALU") / g("SQ_WAVE_CYCLES"),
              "valu_insts_per_dispatch": g("SQ_INSTS_VALU"),      # (the table of tools/sq_summary.py is already per dispatch)
              "l2_hit": hit / (hit + miss) if hit + miss > 0 else None}
import sys as _s, os as _o
_s.path.insert(0, _o.path.dirname(_o.path.dirname(_o.path.abspath(__file__))))
from bench import csrc_sha256
json.dump({"csrc_sha256": csrc_sha256(), "source": src + " (rocprofv3 --pmc SQ_* / GRBM_GUI_ACTIVE passes of one serial bench pass) " + note,
           "definition": "valu_busy_chip = SQ_ACTIVE_INST_VALU x 4 / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs): fraction of the chip's VALU issue "
                         "cycles in use while the kernel runs (every VALU instruction of these kernels is fp64 or integer address arithmetic); "
                         "wave_*: fractions of SQ_WAVE_CYCLES",
           "kernels": ker}, open(out, "w"), indent=1)
for k, v in ker.items():
    print("%-28s valu_busy_chip %.3f  wait %.3f  issue_stall %.3f  wave_valu %.3f" % (k, v["valu_busy_chip"], v["wave_wait_frac"], v["wave_issue_stall_frac"], v["wave_valu_frac"]))
