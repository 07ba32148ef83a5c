#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/final6
mkdir -p $O
cd $R
T0=$(date +%s)
timeout 3000 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$? wall=$(( $(date +%s) - T0 )) s"; tail -3 $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
T0=$(date +%s)
timeout 1200 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default bench rc=$? wall=$(( $(date +%s) - T0 )) s"
python - <<PY
import json
d=json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
print("legs", {k:(v[0], v[1], v[3], v[4], v[5]) for k,v in d["legs_summary"].items()})
print(d["legs"]["msbfs_cross_rmat22"]["first_call"]["first_call_ms_all"], d["first_call"] if "first_call" in d else d["legs"]["prepass"].get("first_call",{}).get("first_call_ms_all"))
PY
timeout 600 python bench.py --workload rmat22_cross --no-cpu-baseline --steps 6 > $O/bench_rmat22_cross.json 2>/dev/null; cut -c1-200 $O/bench_rmat22_cross.json
timeout 600 python bench.py --workload rmat22 --no-cpu-baseline --steps 20 > $O/bench_rmat22.json 2>/dev/null; cut -c1-200 $O/bench_rmat22.json
