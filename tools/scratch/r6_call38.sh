#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6c38
mkdir -p $O
cd $R
timeout 900 python bench.py --config-legs rmat22 --cheapest-pairs 0 --no-cpu-baseline --steps 4 --trace-steps --first-call-handles 4 > $O/a.json 2> $O/a.err
grep "rmat22" $O/a.err | tail -12
python - <<PY
import json
d=json.loads(open("$O/a.json").read().strip().splitlines()[-1])
print("legs", {k:(v[0], v[5]) for k,v in d["legs_summary"].items()})
print(d["legs"]["msbfs_cross_rmat22"].get("first_call"))
PY
