#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6c26
mkdir -p $O
cd $R
PGQ_MEET_TRACE=1 timeout 600 python bench.py --workload rmat22_cross --no-cpu-baseline --no-first-call --steps 2 --warmup 1 --set meet_trace=1 > $O/t.json 2> $O/t.err; grep -i "trace" $O/t.err | tail -6
python - <<PY
import json
d=json.loads(open("$O/t.json").read().strip().splitlines()[-1])
print({k:d.get(k) for k in ("ms_per_step","rows_answered_by_prepass_per_step")})
st=d.get("stats") or {}
print({k:v for k,v in d.items() if k.startswith("ball") or "open" in k})
PY
