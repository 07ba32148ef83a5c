#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6c62
mkdir -p $O
cd $R
for k in 1 2; do
timeout 600 python bench.py --workload rmat22 --no-cpu-baseline --no-first-call --steps 30 > $O/rmat$k.json 2>/dev/null
python - <<PY
import json
d=json.loads(open("$O/rmat$k.json").read().strip().splitlines()[-1])
print("rmat22", round(d["ms_per_step"],4), {k:round(v["ms_per_step"],4) for k,v in d["roofline_by_kernel"].items()})
PY
done
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bibfs or bidirectional or meet_prepass" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
