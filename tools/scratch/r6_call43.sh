#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/pass6
mkdir -p $O
cd $R
T0=$(date +%s)
timeout 1200 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default bench (no flags) rc=$? wall=$(( $(date +%s) - T0 )) s"
python - <<PY
import json
d=json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
print("legs", {k:(v[0], v[1], v[3], v[4], v[5]) for k,v in d["legs_summary"].items()})
print(d["roofline"]["frac"], d["cpu_baseline"]["rows_equal"])
PY
tail -3 $O/bench_default.err
