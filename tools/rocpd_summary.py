"""Turn a rocprofv3 rocpd database (ROCm 7.2 default output) into the per-kernel stats table that
`--stats` used to print as CSV:  python tools/rocpd_summary.py <results.db> [out.csv]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, total_calls, total_duration, average, percentage from top_kernels "
                  "order by total_duration desc").fetchall()
lines = ["Name,Calls,TotalDurationNs,AverageNs,Percentage"]
for n, c, t, a, p in rows:
    lines.append('"%s",%d,%d,%.1f,%.2f' % (n.split("(")[0], c, t, a, p))
regs = db.execute("select distinct name, vgpr_count, accum_vgpr_count, sgpr_count, lds_size, scratch_size, "
                  "grid_x, grid_y, grid_z, workgroup_x from kernels group by name").fetchall()
lines.append("")
lines.append("Name,VGPR,AGPR,SGPR,LDS,Scratch,GridX,GridY,GridZ,WorkgroupX")
for r in regs:
    lines.append('"%s",%s' % (r[0].split("(")[0], ",".join(str(x) for x in r[1:])))
out = "\n".join(lines) + "\n"
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(out)
print(out)
