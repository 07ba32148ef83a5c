#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6c13
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "ball or cheapest or first_call" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --no-cpu-baseline --no-first-call --steps 10 --warmup 2 $XA > $O/$tag.json 2> $O/$tag.err
  python - <<PY
import json
try:
    d=json.load(open("$O/$tag.json")); k=d["roofline_by_kernel"]
    print("$tag", "ms/step %.4f"%d["ms_per_step"], {n:k[n]["ms_per_step"] for n in k}, "chain frac %.3f step frac %.3f"%(d["roofline"]["frac"], d["roofline"]["step"]["frac"]))
except Exception as e: print("$tag", "failed", e)
PY
}
XA="--workload snb_cross"
run x A=1
run x_nohead PGQ_BALL_HEAD_MB=0
XA="--workload snb_cross --cross-dests 128 --pairs-per-gpu 262144"
run x128 A=1
XA="--workload snb_cheapest --steps 1 --warmup 1"
run cheapest A=1
bash tools/prof_quick.sh r6c13/prof_cross pmc --workload snb_cross 2>&1 | grep -i "k_src_ball\|k_ball_seg" | head -6
