#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6c8
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "ball" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
run() { tag=$1; shift; env "$@" timeout 400 python bench.py --no-cpu-baseline --steps 10 $XA > $O/$tag.json 2> $O/$tag.err
  python - <<PY
import json
try:
    d=json.load(open("$O/$tag.json")); k=d["roofline_by_kernel"]
    print("$tag", "ms/step %.4f"%d["ms_per_step"], {n:k[n]["ms_per_step"] for n in k}, "chain frac %.3f step frac %.3f"%(d["roofline"]["frac"], d["roofline"]["step"]["frac"]))
except Exception as e: print("$tag", "failed", e)
PY
  grep "k_src_ball trace" $O/$tag.err | tail -1; }
XA="--workload snb_cross"
run x A=1
XA="--workload snb_cross --cross-dests 32 --pairs-per-gpu 65536"
run x32 A=1
run x32_forced PGQ_BALL=2
XA="--workload snb_cross --cross-dests 128 --pairs-per-gpu 262144"
run x128 A=1
XA="--workload snb_cross_allv --steps 3"
run allv A=1
XA="--no-legs --steps 20 --warmup 5"
run default A=1
run default_noball PGQ_BALL=0
XA="--no-legs --pairs-per-gpu 8192"
run p8192 A=1
run p8192_noball PGQ_BALL=0
