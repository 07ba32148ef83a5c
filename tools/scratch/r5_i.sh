#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r5i
mkdir -p $O
cd $R
short() { python - "$1" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    r=json.loads(l); print(r["tag"], r["cfg"] or "default", r["n"], "wall", r["wall_ms"], "same", r["same_as_first"], {k:(v["ms"]) for k,v in r["kernels"].items()})
PY
}
rm -f $O/*.jsonl
for x in "65536 2048" "262144 2048" "524288 2048" "1048576 2048" "2097152 2048" "65536 64" "1000000 100000"; do
  set -- $x
  timeout 600 python tools/sweep_meet.py --pairs $1 --cross $2 --tag x$2_$1 --out $O/x.jsonl --steps 5 --configs ";meet_bias=100;meet=0" > /dev/null 2>&1
done
short $O/x.jsonl
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "route_memo or prepass or sf100 or 65536 or cross_product" 2>&1 | tail -3
