#!/bin/bash
# kernel resource table of one HIP source (VGPRs, AGPRs, spills, scratch, LDS, occupancy):  tools/rsrc.sh lf_front.hip [extra flags]
src=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -c /root/repo/lineslam_amd/csrc/$src -o /tmp/rsrc_$$.o "$@" \
  -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import sys,re
cur={}
rows=[]
for l in sys.stdin:
    if "error:" in l: print(l,end="")
    m=re.search(r"remark:\s+([A-Za-z][\w ]*?)(?: \[[^\]]*\])?:\s+(\S+)",l)
    if not m: continue
    k,v=m.group(1).strip(),m.group(2)
    if k=="Function Name":
        cur={"name":v}; rows.append(cur)
    else: cur[k]=v
print("%-58s %5s %5s %5s %6s %6s %7s %7s %4s"%("kernel","SGPR","VGPR","AGPR","vspill","sspill","scratch","LDS","occ"))
for r in rows:
    print("%-58s %5s %5s %5s %6s %6s %7s %7s %4s"%(r["name"][:58],r.get("TotalSGPRs"),r.get("VGPRs"),r.get("AGPRs"),r.get("VGPRs Spill"),r.get("SGPRs Spill"),r.get("ScratchSize"),r.get("LDS Size"),r.get("Occupancy")))
'
rm -f /tmp/rsrc_$$.o
