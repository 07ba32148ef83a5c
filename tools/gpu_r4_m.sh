#!/bin/bash
# round 4, call M: stage A of k_meet4d's distance-4 step (both prefixes in one round trip)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r4n
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "meet or bibfs or c2_rmat or unpinned or large_inputs or golden or full_size or fuzz" > $O/pytest_sub.txt 2>&1; tail -3 $O/pytest_sub.txt
S="python tools/sweep_meet.py --steps 20 --out $O/sweep.jsonl"
for n in 65536 8192 2048; do timeout 200 $S --tag b$n --pairs $n > /dev/null 2>&1; done
python - <<PY
import json
for l in open("$O/sweep.jsonl"):
    r=json.loads(l); print(r["tag"], "wall", r["wall_ms"], "same", r["same_as_first"], {k:v for k,v in r["kernels"].items() if k in ("meet","meet4","bibfs")})
PY
PGQ_MEET_TRACE=1 timeout 200 $S --tag trace --pairs 65536 --steps 3 2>&1 >/dev/null | grep "k_meet4d trace" | tail -1
timeout 300 python bench.py --workload rmat22 --no-cpu-baseline > $O/b_rmat.json 2>/dev/null; python - <<PY
import json
j=json.load(open("$O/b_rmat.json")); print("rmat22 ms", round(j["ms_per_step"],4), {k:v["ms_per_step"] for k,v in j["roofline_by_kernel"].items()})
PY
