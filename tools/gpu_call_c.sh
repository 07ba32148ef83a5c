#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
O=gpurun_out/r3c; mkdir -p $O; rm -f $O/sweep.jsonl
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "golden or fuzz or unpinned or random_graph or shared_sources or rmat18 or literal or large_inputs" > $O/pytest_meet.log 2>&1; tail -3 $O/pytest_meet.log
S="python tools/sweep_meet.py --out $O/sweep.jsonl"
P='import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d["tag"], d["cfg"], d["n"], d["wall_ms"], {k:(v["ms"],v["GBps"]) for k,v in d["kernels"].items()}, d["edges_scanned"], d["meet_pairs"], d["levels"], d["same_as_first"])'
timeout 300 $S --tag cross64k --cross 2048 --steps 3 --configs ";probe=0" 2> /dev/null | python -c "$P"
timeout 300 $S --tag cross2M --cross 2048 --pairs 2097152 --steps 3 --configs "trace=1;probe=0" 2> $O/sweep_cross.err | python -c "$P"
grep "batch" $O/sweep_cross.err | tail -6 | cut -c1-200
timeout 300 $S --tag cross14M --cross 32 --pairs 14356032 --steps 2 --configs "" 2> /dev/null | python -c "$P"
