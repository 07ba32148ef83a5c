#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r5m
mkdir -p $O
cd $R
run() { env "$@" timeout 300 python bench.py --workload rmat22 --no-cpu-baseline --steps 10 > $O/b.json 2>$O/b.err
python - "$*" <<'PY'
import json,sys
o=json.load(open("/root/repo/gpurun_out/r5m/b.json")); print(sys.argv[1], round(o["ms_per_step"],4), {k:v["ms_per_step"] for k,v in o["roofline_by_kernel"].items()}, o["rows_answered_by_prepass_per_step"], o["levels_per_step"])
PY
}
run PGQ_TRACE=0
run PGQ_BIBFS_GRID=256
run PGQ_BIBFS_GRID=128
run PGQ_MEET_TRACE=1
tail -5 $O/b.err
