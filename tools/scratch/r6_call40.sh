#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/pass6
mkdir -p $O
cd $R
T0=$(date +%s)
timeout 900 python tools/chunk_latency.py > $O/chunk_latency.json 2> $O/chunk_latency.err; echo "chunk_latency rc=$? wall=$(( $(date +%s) - T0 )) s"; cat $O/chunk_latency.json; tail -3 $O/chunk_latency.err
T0=$(date +%s)
timeout 1200 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "default bench rc=$? wall=$(( $(date +%s) - T0 )) s"
python - <<PY
import json
d=json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
print("legs", {k:(v[0], v[1], v[5]) for k,v in d["legs_summary"].items()})
PY
