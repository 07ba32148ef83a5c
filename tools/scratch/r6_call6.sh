#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6c6
mkdir -p $O
cd $R
run() { tag=$1; shift; env "$@" timeout 400 python bench.py --workload snb_cross --no-cpu-baseline --steps 5 $XA > $O/$tag.json 2> $O/$tag.err
  python - <<PY
import json
try:
    d=json.load(open("$O/$tag.json")); k=d["roofline_by_kernel"]
    print("$tag", "ms/step %.4f"%d["ms_per_step"], {n:k[n]["ms_per_step"] for n in k})
except Exception as e: print("$tag", "failed", e)
PY
  grep "k_src_ball trace" $O/$tag.err | tail -1; }
XA="--cross-dests 32 --pairs-per-gpu 65536"
run x32_trace PGQ_MEET_TRACE=1
run x32_tcap64 PGQ_BALL_TEST_CAP=64
run x32_grid256 PGQ_BALL_GRID=256
run x32_grid128 PGQ_BALL_GRID=128
XA=""
run x_grid256 PGQ_BALL_GRID=256
run x_grid128 PGQ_BALL_GRID=128
run x_grid512 PGQ_BALL_GRID=512
