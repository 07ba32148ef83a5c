#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r5l
mkdir -p $O
cd $R
run() { env "$@" timeout 300 python bench.py --workload snb_cross --no-cpu-baseline --steps 10 > $O/b.json 2>/dev/null
python - "$*" <<'PY'
import json,sys
o=json.load(open("/root/repo/gpurun_out/r5l/b.json")); print(sys.argv[1], round(o["ms_per_step"],4), {k:v["ms_per_step"] for k,v in o["roofline_by_kernel"].items()})
PY
}
run PGQ_TRACE=0
run PGQ_PUSH_CHUNK=64
run PGQ_PUSH_CHUNK=128
run PGQ_BLOCKS_PER_CU=6
run PGQ_BLOCKS_PER_CU=10
run PGQ_LANES_UNROLL=4
run PGQ_LANES_UNROLL=1
run PGQ_TRACE=0
