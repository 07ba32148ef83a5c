#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6c15
mkdir -p $O
cd $R
for v in base w4 w4h8 h2 h8; do
  L=""; [ $v != base ] && L=$R/build_variants/libpgq_hip_$v.so
  PGQ_HIP_LIB=$L timeout 400 python bench.py --workload snb_cross --no-cpu-baseline --no-first-call --steps 10 > $O/x_$v.json 2>/dev/null
  PGQ_HIP_LIB=$L timeout 400 python bench.py --workload snb_cross --cross-dests 128 --pairs-per-gpu 262144 --no-cpu-baseline --no-first-call > $O/x128_$v.json 2>/dev/null
  python - <<PY
import json
for f in ("x_$v","x128_$v"):
    try:
        d=json.load(open("$O/%s.json"%f)); k=d["roofline_by_kernel"]
        print("$v", f, "ms/step %.4f"%d["ms_per_step"], {n:k[n]["ms_per_step"] for n in k})
    except Exception as e: print("$v", f, "failed", e)
PY
done
timeout 300 python tools/chunk_throughput.py > $O/chunk_throughput.json 2> $O/chunk_throughput.err; cat $O/chunk_throughput.json; tail -2 $O/chunk_throughput.err
