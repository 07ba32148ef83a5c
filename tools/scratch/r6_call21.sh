#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6c21
mkdir -p $O
cd $R
run() { tag=$1; shift; timeout 600 python bench.py --workload snb_cheapest --no-cpu-baseline --no-first-call --steps 2 --warmup 1 "$@" > $O/$tag.json 2> $O/$tag.err; python - <<PY
import json
d=json.loads(open("$O/$tag.json").read().strip().splitlines()[-1])
print("$tag", "ms", round(d["ms_per_step"],2), "prepass rows", d.get("rows_answered_by_prepass_per_step"), "levels", d.get("levels_per_step"), "batches", d.get("batches_per_step"))
PY
}
run base --pairs-per-gpu 512
run wb64 --pairs-per-gpu 512 --set wbibfs=1
run wb8 --pairs-per-gpu 512 --set wbibfs=1 --set wbibfs_delta_div=8
run wb8p --pairs-per-gpu 512 --set wbibfs=1 --set wbibfs_delta_div=8 --set wbibfs_prune=1
run wb2p --pairs-per-gpu 512 --set wbibfs=1 --set wbibfs_delta_div=2 --set wbibfs_prune=1
