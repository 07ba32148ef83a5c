#!/bin/bash
# tools/kres.sh file.hip [extra hipcc flags]: per-kernel registers / scratch / occupancy of a gfx950 build (no GPU needed)
cd "$(dirname "$0")/../duckpgq-extension_amd/csrc" || exit 1
f=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -I. -DNDEBUG "$@" -c -o /tmp/kres_$$.o "$f" \
  -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import re,sys,subprocess
cur=None;rows=[]
for l in sys.stdin:
    m=re.search(r"Function Name: (\S+)",l)
    if m:
        cur={"name":m.group(1)};rows.append(cur);continue
    m=re.search(r"remark:\s+(\w[\w ]*?)(?: \[[^\]]*\])?: (\d+) \[-Rpass",l)
    if m and cur is not None: cur[m.group(1).strip()]=int(m.group(2))
rows=[r for r in rows if "rocprim" not in r["name"] and "hipcub" not in r["name"]]
names=subprocess.run(["c++filt"]+[r["name"] for r in rows],capture_output=True,text=True).stdout.split("\n")
for r,n in zip(rows,names):
    n=re.sub(r"\(.*","",n)
    print("%-44s vgpr %3d agpr %3d sgpr %3d scratch %4d occ %2d lds %6d sspill %3d vspill %3d"%(n[:44],r.get("VGPRs",-1),r.get("AGPRs",-1),r.get("TotalSGPRs",-1),r.get("ScratchSize",-1),r.get("Occupancy",-1),r.get("LDS Size",-1),r.get("SGPRs Spill",-1),r.get("VGPRs Spill",-1)))
'
rm -f /tmp/kres_$$.o
