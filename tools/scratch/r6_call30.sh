#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6c30
mkdir -p $O
cd $R
run() { wl=$1; tag=$2; shift; shift; timeout 600 python bench.py --workload $wl --no-cpu-baseline --no-first-call --steps 20 "$@" > $O/$tag.json 2> $O/$tag.err; python - <<PY
import json
try:
    d=json.loads(open("$O/$tag.json").read().strip().splitlines()[-1])
    print("$tag", "ms", round(d["ms_per_step"],4), {k:round(v["ms_per_step"],4) for k,v in d["roofline_by_kernel"].items()})
except Exception as e:
    print("$tag failed", e); print(open("$O/$tag.err").read()[-800:])
PY
}
run rmat22 rmat
run rmat22 rmat_trace --set meet_trace=1
grep "trace" $O/rmat_trace.err | tail -2
run snb_sf100 snb --no-legs
run snb_sf100 snb8192 --no-legs --pairs-per-gpu 8192
