#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r5f
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "enqueued_ahead or route_memo or zero_copy or ring_with_chords or cheapest or weighted or two_ranks or prepass or sf100 or c4 or 65536" > $O/pytest_f.txt 2>&1; tail -15 $O/pytest_f.txt
for i in 1 2; do
timeout 300 python bench.py --no-legs --no-cpu-baseline --steps 20 --warmup 5 > $O/b$i.json 2>/dev/null
python - <<PY
import json
o=json.load(open("$O/b$i.json")); print(round(o["ms_per_step"],4), o["roofline"]["frac"], o["roofline"]["step"]["frac"], o["roofline_by_kernel"])
PY
done
