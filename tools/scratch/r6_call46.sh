#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6c46
mkdir -p $O
cd $R
for q in default 8 16 24; do
  if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  timeout 300 python tools/chunk_throughput.py > $O/ct_$q.txt 2> $O/ct_$q.err
  python - <<PY
import json
d=json.loads(open("$O/ct_$q.txt").read().strip().splitlines()[-1])
print("$q", {k.replace("cross_1src_x_2048dst_","x").replace("scattered_2048_pairs_","s"): round(v["rows_per_s"]/1e6,1) for k,v in d.items()})
PY
done
