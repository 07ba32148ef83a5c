#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
O=gpurun_out/r3c; mkdir -p $O; rm -f $O/sweep.jsonl
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "meet or golden or bibfs or fuzz or unpinned" > $O/pytest_meet.log 2>&1; tail -3 $O/pytest_meet.log
S="python tools/sweep_meet.py --out $O/sweep.jsonl"
P='import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d["tag"], d["cfg"], d["wall_ms"], d["kernels"], d["edges_scanned"], d["meet_pairs"], d["levels"])'
timeout 300 $S --tag base --configs ";meet4_grid_mult=2" 2> $O/sweep.err | python -c "$P"
for v in md4 md4d4 b5md4d4 w6; do
	PGQ_HIP_LIB=$R/build_variants/libpgq_hip_$v.so timeout 120 $S --tag $v --configs "" 2>> $O/sweep.err | python -c "$P"
done
