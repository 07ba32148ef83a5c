#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6c50
mkdir -p $O
cd $R
run() { tag=$1; shift; timeout 600 env "$@" > $O/$tag.json 2> $O/$tag.err; python - <<PY
import json
try:
    d=json.loads(open("$O/$tag.json").read().strip().splitlines()[-1])
    print("$tag", "ms", round(d["ms_per_step"],4), "levels", d["levels_per_step"], "deferred", d["deferred_pairs_per_step"], {k:round(v["ms_per_step"],3) for k,v in d["roofline_by_kernel"].items()}, "frac", round(d["roofline"]["frac"],3), round(d["roofline"]["step"]["frac"],3))
except Exception as e:
    print("$tag failed", e); print(open("$O/$tag.err").read()[-800:])
PY
}
run rmatx_lanes_noprobe2 PGQ_MEET=0 PGQ_PROBE2=0 python bench.py --workload rmat22_cross --no-cpu-baseline --no-first-call --steps 4 --warmup 2
run rmatx_lanes_p2_512 PGQ_MEET=0 PGQ_PROBE2_ABS=512 python bench.py --workload rmat22_cross --no-cpu-baseline --no-first-call --steps 4 --warmup 2
run snbx_lanes_noprobe2 PGQ_BALL=0 PGQ_PROBE2=0 python bench.py --workload snb_cross --no-cpu-baseline --no-first-call --steps 6
