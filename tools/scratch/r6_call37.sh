#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6c37
mkdir -p $O
cd $R
timeout 900 python bench.py --config-legs rmat22 --cheapest-pairs 0 --no-cpu-baseline --no-first-call --steps 4 > $O/a.json 2> $O/a.err
python - <<PY
import json
d=json.loads(open("$O/a.json").read().strip().splitlines()[-1])
print("legs", {k:v[0] for k,v in d["legs_summary"].items()})
PY
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --hip-trace --stats --output-format csv -d $O/hip -o h -- python $R/bench.py --config-legs rmat22 --cheapest-pairs 0 --no-cpu-baseline --no-first-call --steps 4 > $O/b.json 2> $O/b.err
rm -f $O/hip/*hip_api_trace.csv
python - <<PY
import csv,glob
for p in glob.glob("$O/hip/*hip_api_stats.csv"):
    rows=list(csv.DictReader(open(p)))
    for r in rows[:12]:
        print("%-40s calls %7s total_ms %10.1f avg_us %9.1f" % (r["Name"][:40], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3))
PY
