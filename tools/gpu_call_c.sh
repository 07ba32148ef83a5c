#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
O=gpurun_out/r3c; mkdir -p $O; rm -f $O/sweep.jsonl
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "meet or golden or bibfs or fuzz or unpinned" > $O/pytest_meet.log 2>&1; tail -3 $O/pytest_meet.log
S="python tools/sweep_meet.py --out $O/sweep.jsonl"
P='import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d["tag"], d["cfg"], d["wall_ms"], d["kernels"].get("meet"), d["edges_scanned"], d["meet_pairs"], d["same_as_first"])'
timeout 300 $S --tag base --configs "meet_align=16;meet_align=16,meet4=0;meet_align=32" 2> $O/sweep.err | python -c "$P"
for v in w6 w6d3; do
	PGQ_HIP_LIB=$R/build_variants/libpgq_hip_$v.so timeout 120 $S --tag $v --configs "meet_align=16;meet_align=16,meet4=0" 2>> $O/sweep.err | python -c "$P"
done
bash tools/prof_meet.sh r3c/prof 2>&1 | grep -E "k_meet3|k_meet4d" | cut -c1-1500
