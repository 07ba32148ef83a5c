#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r5c
mkdir -p $O
cd $R
for v in "4 8" "4 4" "4 2" "2 8" "1 8"; do
  set -- $v
  PGQ_DETECT_UNROLL=$1 PGQ_DETECT_GRID_MULT=$2 timeout 300 python bench.py --workload snb_cross --no-cpu-baseline --steps 10 > $O/b.json 2>/dev/null
  python - <<PY
import json
o=json.load(open("$O/b.json")); print("unroll $1 grid_mult $2", round(o["ms_per_step"],4), o["roofline_by_kernel"]["detect"])
PY
done
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python $R/bench.py --workload snb_cross --no-cpu-baseline --steps 3 --warmup 1 > $O/trace.log 2>&1)
python - <<PY
import csv,glob
f=glob.glob("$O/trace/*kernel_trace.csv")[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
prev=None
out=[]
for r in rows:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    out.append((r["Kernel_Name"][:60],(e-s)/1e3,(s-prev)/1e3 if prev else 0))
    prev=e
open("$O/dispatches.txt","w").write("\n".join("%-60s %9.1f us gap %9.1f"%x for x in out))
PY
rm -rf $O/trace
grep -n "k_scatter_results" $O/dispatches.txt | tail -3
