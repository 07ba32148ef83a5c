#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6c29
mkdir -p $O
cd $R
run() { tag=$1; shift; timeout 600 python bench.py --workload rmat22 --no-cpu-baseline --no-first-call --steps 20 "$@" > $O/$tag.json 2> $O/$tag.err; python - <<PY
import json
try:
    d=json.loads(open("$O/$tag.json").read().strip().splitlines()[-1])
    print("$tag", "ms", round(d["ms_per_step"],4), {k:round(v["ms_per_step"],4) for k,v in d["roofline_by_kernel"].items()})
except Exception as e:
    print("$tag failed", e); print(open("$O/$tag.err").read()[-800:])
PY
}
run base
run cap8k --set meet_cap_small=8192
run cap4k --set meet_cap_small=4096
run cap2k --set meet_cap_small=2048
run cap4k_t8k --set meet_cap_small=4096 --set meet4_test_cap=8192
run cap4k_m4cap --set meet_cap_small=4096 --set meet4_cap=262144
