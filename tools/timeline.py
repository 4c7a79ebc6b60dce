"""python tools/timeline.py <kernel_trace.csv> [t0_ms t1_ms]: a text Gantt of the kernels of a rocprofv3 --kernel-trace run
inside a window (default: the last 400 ms), one row per queue, merged per kernel name -- shows what overlaps with what."""
import csv, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
print("columns:", list(rows[0].keys()))
key = "Stream_Id" if "Stream_Id" in rows[0] else "Queue_Id"
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("<")[0], r.get(key, "0")) for r in rows]
tmax = max(e[1] for e in ev)
t1 = tmax - (float(sys.argv[3]) * 1e6 if len(sys.argv) > 3 else 0)
t0 = t1 - (float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 400e6)
ev = [e for e in ev if e[1] > t0 and e[0] < t1]
qs = sorted(set(e[3] for e in ev))
W = 160
dt = (t1 - t0) / W
sym = {}
def s(name):
    if name not in sym:
        sym[name] = "abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ"[len(sym) % 52]
    return sym[name]
for q in qs:
    line = [" "] * W
    for a, b, n, qq in ev:
        if qq != q:
            continue
        for c in range(max(0, int((a - t0) / dt)), min(W, int((b - t0) / dt) + 1)):
            line[c] = s(n)
    print("q%-3s|%s|" % (q, "".join(line)))
print("window %.1f ms, %.2f ms per column" % ((t1 - t0) / 1e6, dt / 1e6))
pts = sorted([(max(a, t0), 1) for a, b, n, q in ev] + [(min(b, t1), -1) for a, b, n, q in ev])
busy = 0.0; depth = 0; last = t0; area = 0.0
for t, d in pts:
    if depth > 0: busy += t - last
    area += depth * (t - last); last = t; depth += d
print("GPU busy %.1f ms of the window; mean kernels in flight while busy %.2f" % (busy / 1e6, area / max(busy, 1)))
tot = defaultdict(float)
for a, b, n, q in ev:
    tot[n] += (min(b, t1) - max(a, t0)) / 1e6
for n in sorted(tot, key=lambda k: -tot[k])[:16]:
    print(" %s %-28s %8.1f ms in window" % (s(n), n, tot[n]))
