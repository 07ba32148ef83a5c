#!/bin/bash
# round 4, call L: fewer wavefronts per k_meet4d row (a row's phases are bursts of redundant control work of all its wavefronts)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r4l
mkdir -p $O
cd $R
S="python tools/sweep_meet.py --steps 20 --out $O/sweep.jsonl"
for v in base w8 w8d4 w4d4; do
	for n in 65536 8192 2048; do
		if [ $v = base ]; then timeout 200 $S --tag ${v}_$n --pairs $n > /dev/null 2>&1; else PGQ_HIP_LIB=$R/build_variants/libpgq_hip_$v.so timeout 200 $S --tag ${v}_$n --pairs $n > /dev/null 2>&1; fi
	done
done
python - <<PY
import json
for l in open("$O/sweep.jsonl"):
    r=json.loads(l); print(r["tag"], "wall", r["wall_ms"], "same", r["same_as_first"], {k:v["ms"] for k,v in r["kernels"].items() if k in ("meet","meet4","bibfs")})
PY
for v in w8 w4d4; do PGQ_HIP_LIB=$R/build_variants/libpgq_hip_$v.so timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "meet_prepass or bibfs or unpinned" 2>&1 | tail -1; done
