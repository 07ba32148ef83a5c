#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6c56
mkdir -p $O
cd $R
for sr in 16384 4096; do
  for n in 4096 6144 8192 12288 16384; do
    timeout 300 python bench.py --no-legs --no-cpu-baseline --no-first-call --pairs-per-gpu $n --steps 40 --warmup 5 --set meet_small_rows=$sr > $O/b_${sr}_$n.json 2>/dev/null
    python - <<PY
import json
d=json.loads(open("$O/b_${sr}_$n.json").read().strip().splitlines()[-1])
print("small_rows=$sr n=$n ms", round(d["ms_per_step"],4), {k:round(v["ms_per_step"],4) for k,v in d["roofline_by_kernel"].items()})
PY
  done
done
