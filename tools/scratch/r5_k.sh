#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r5k
mkdir -p $O
cd $R
for i in 1 2; do
timeout 300 python bench.py --workload snb_cross --no-cpu-baseline --steps 10 > $O/b$i.json 2>/dev/null
python - <<PY
import json
o=json.load(open("$O/b$i.json")); print(round(o["ms_per_step"],4), o["roofline"]["step"]["frac"], {k:v["ms_per_step"] for k,v in o["roofline_by_kernel"].items()})
PY
done
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "random_graph_all_variants or cross_product or enqueued or fuzz or sf100 or probe" 2>&1 | tail -3
