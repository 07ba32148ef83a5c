#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6c39
mkdir -p $O
cd $R
timeout 900 python bench.py --no-cpu-baseline --steps 4 --trace-steps > $O/a.json 2> $O/a.err
grep "rmat22_cross\|cheapest" $O/a.err | tail -12
python - <<PY
import json
d=json.loads(open("$O/a.json").read().strip().splitlines()[-1])
print("legs", {k:(v[0], v[5]) for k,v in d["legs_summary"].items()})
PY
timeout 900 python bench.py --no-cpu-baseline --steps 4 --config-legs rmat22 > $O/b.json 2> $O/b.err
python - <<PY
import json
d=json.loads(open("$O/b.json").read().strip().splitlines()[-1])
print("legs b (cheapest then rmat22)", {k:(v[0], v[5]) for k,v in d["legs_summary"].items()})
PY
