#!/bin/bash
# round 4, call J: R-MAT-22 sensitivity to the caps
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r4j
mkdir -p $O
cd $R
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --workload rmat22 --no-cpu-baseline > $O/b_$tag.json 2>/dev/null; python - <<PY
import json
j=json.load(open("$O/b_$tag.json")); print("$tag", "$*", "ms", round(j["ms_per_step"],4), {k:v["ms_per_step"] for k,v in j["roofline_by_kernel"].items()}, "levels", j["levels_per_step"], "prepass rows", j["rows_answered_by_prepass_per_step"])
PY
}
run a PGQ_MEET4_TEST_CAP=32768
run b PGQ_MEET4_TEST_CAP=32768 PGQ_MEET4_CAP=262144
run c PGQ_MEET4_TEST_CAP=32768 PGQ_MEET4_CAP=262144 PGQ_MEET_CAP_SMALL=4096
run d PGQ_MEET4_TEST_CAP=8192 PGQ_MEET4_CAP=131072 PGQ_MEET_CAP_SMALL=4096
run e PGQ_MEET4_TEST_CAP=32768 PGQ_MEET4_CAP=262144 PGQ_BIBFS_CAP=1048576
