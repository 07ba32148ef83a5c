#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6c10
mkdir -p $O
cd $R
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --no-cpu-baseline --no-first-call --steps 4 --warmup 1 $XA > $O/$tag.json 2> $O/$tag.err
  python - <<PY
import json
try:
    d=json.load(open("$O/$tag.json")); k=d["roofline_by_kernel"]
    print("$tag", "ms/step %.4f"%d["ms_per_step"], {n:k[n]["ms_per_step"] for n in k}, "chain frac %.3f step frac %.3f"%(d["roofline"]["frac"], d["roofline"]["step"]["frac"]), "levels", d["levels_per_step"], d["push_pull_levels"], "prepass rows", d["rows_answered_by_prepass_per_step"])
except Exception as e: print("$tag", "failed", e)
PY
  tail -2 $O/$tag.err | cut -c1-300; }
XA="--workload rmat22_cross"
run rx_ball A=1
run rx_noball PGQ_BALL=0
run rx_nomeet PGQ_MEET=0
run rx_noball_trace PGQ_BALL=0 PGQ_TRACE=1
grep "\[pgq\] batch" $O/rx_noball_trace.err | tail -40 | cut -c1-220
