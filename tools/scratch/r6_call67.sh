#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6c67
mkdir -p $O
cd $R
for v in base static base static; do
  if [ $v = static ]; then export PGQ_HIP_LIB=$R/build_variants/libpgq_hip_static.so; else unset PGQ_HIP_LIB; fi
  timeout 300 python bench.py --workload snb_cross --no-cpu-baseline --no-first-call --steps 20 > $O/$v.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("$O/$v.json").read().strip().splitlines()[-1])
print("$v", round(d["ms_per_step"],4), {k:round(v["ms_per_step"],4) for k,v in d["roofline_by_kernel"].items()})
PY
done
