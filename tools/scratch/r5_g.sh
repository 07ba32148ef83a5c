#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r5g
mkdir -p $O
cd $R
short() { python - "$1" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    r=json.loads(l); print(r["tag"], r["cfg"] or "default", r["n"], "wall", r["wall_ms"], "same", r["same_as_first"], {k:(v["ms"]) for k,v in r["kernels"].items()})
PY
}
rm -f $O/*.jsonl
timeout 600 python tools/sweep_meet.py --pairs 65536 --tag p65536 --out $O/a.jsonl --configs ";meet_cap=8192;meet_cap=32768;meet4_grid_mult=1;meet4_grid_mult=3;meet4_test_cap=16384;meet_grid_mult=4;meet_grid_mult=16" > /dev/null 2>&1; short $O/a.jsonl
timeout 600 python tools/sweep_meet.py --pairs 8192 --tag p8192 --out $O/b.jsonl --configs ";meet_cap_small=4096;meet_cap_small=2048;meet_cap_small=1024" > /dev/null 2>&1; short $O/b.jsonl
timeout 600 python tools/sweep_meet.py --pairs 2048 --tag p2048 --out $O/c.jsonl --configs ";meet_cap_small=4096;meet_cap_small=1024" > /dev/null 2>&1; short $O/c.jsonl
timeout 600 python tools/sweep_meet.py --pairs 65536 --cross 2048 --tag x2048x32 --out $O/d.jsonl --configs ";meet_bias=100;spec_levels=0;probe=0" > /dev/null 2>&1; short $O/d.jsonl
timeout 600 python tools/sweep_meet.py --pairs 8192 --tag p8192_meet0 --out $O/e.jsonl --configs "meet=0;meet=0,spec_levels=0" > /dev/null 2>&1; short $O/e.jsonl
