#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r5h
mkdir -p $O
cd $R
short() { python - "$1" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    r=json.loads(l); print(r["tag"], r["cfg"] or "default", r["n"], "wall", r["wall_ms"], "same", r["same_as_first"], {k:(v["ms"]) for k,v in r["kernels"].items()})
PY
}
rm -f $O/*.jsonl
timeout 600 python tools/sweep_meet.py --pairs 65536 --tag p65536 --out $O/a.jsonl --configs ";meet_cap=8192;meet_cap=12288;meet_cap=24576;meet4_test_cap=16384;meet4_grid_mult=2" > /dev/null 2>&1; short $O/a.jsonl
timeout 600 python tools/sweep_meet.py --pairs 8192 --tag p8192 --out $O/b.jsonl --configs ";meet4_grid_mult=2" > /dev/null 2>&1; short $O/b.jsonl
timeout 600 python tools/sweep_meet.py --pairs 65536 --cross 2048 --tag x2048x32 --out $O/d.jsonl --configs ";meet_calibrate=0" > /dev/null 2>&1; short $O/d.jsonl
timeout 600 python tools/sweep_meet.py --pairs 262144 --cross 2048 --tag x2048x128 --out $O/e.jsonl --configs ";meet_bias=100;meet=0" > /dev/null 2>&1; short $O/e.jsonl
timeout 600 python tools/sweep_meet.py --pairs 524288 --cross 2048 --tag x2048x256 --out $O/f.jsonl --configs ";meet_bias=100;meet=0" > /dev/null 2>&1; short $O/f.jsonl
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "route_memo or prepass or sf100 or 65536 or cross_product" 2>&1 | tail -3
