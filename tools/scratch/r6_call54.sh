#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6c54
mkdir -p $O
cd $R
for sp in 0 1; do
  if [ $sp = 1 ]; then export PGQ_MEET_SPIN=1; else unset PGQ_MEET_SPIN; fi
  for n in 2048 8192 65536; do
    timeout 300 python bench.py --no-legs --no-cpu-baseline --no-first-call --pairs-per-gpu $n --steps 50 --warmup 5 > $O/b_${sp}_$n.json 2>/dev/null
    python - <<PY
import json
d=json.loads(open("$O/b_${sp}_$n.json").read().strip().splitlines()[-1])
print("spin=$sp n=$n ms", round(d["ms_per_step"],4))
PY
  done
  timeout 300 python tools/chunk_latency.py 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('spin=$sp', {k:round(v,4) for k,v in d.items() if 'iterativelength' in k and 'py' not in k})"
done
