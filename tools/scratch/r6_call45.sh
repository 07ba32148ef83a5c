#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6c45
mkdir -p $O
cd $R
timeout 900 python bench.py --config-legs rmat22 --cheapest-pairs 0 --no-cpu-baseline --steps 6 > $O/a.json 2> $O/a.err
python - <<PY
import json
d=json.loads(open("$O/a.json").read().strip().splitlines()[-1])
print("legs", {k:(v[0], v[5]) for k,v in d["legs_summary"].items()})
print(d["legs"]["msbfs_cross_rmat22"].get("first_call",{}).get("first_call_ms_all"))
PY
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bibfs or meet_prepass or first_call" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
