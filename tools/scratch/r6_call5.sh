#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6c5
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "ball" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
for v in base w4 w4u4 u4 u1; do
  L=""; [ $v != base ] && L=$R/build_variants/libpgq_hip_$v.so
  PGQ_HIP_LIB=$L timeout 400 python bench.py --workload snb_cross --no-cpu-baseline --steps 10 > $O/x_$v.json 2>/dev/null
  PGQ_HIP_LIB=$L timeout 400 python bench.py --workload snb_cross --cross-dests 32 --pairs-per-gpu 65536 --no-cpu-baseline > $O/x32_$v.json 2>/dev/null
  python - <<PY
import json
for f in ("x_$v","x32_$v"):
    try:
        d=json.load(open("$O/%s.json"%f)); k=d["roofline_by_kernel"]
        print("$v", f, "ms/step %.4f"%d["ms_per_step"], {n:k[n]["ms_per_step"] for n in k})
    except Exception as e: print("$v", f, "failed", e)
PY
done
PGQ_MEET_TRACE=1 timeout 400 python bench.py --workload snb_cross --no-cpu-baseline --steps 3 --warmup 1 > $O/t1.json 2> $O/t1.err; grep "k_src_ball trace" $O/t1.err | tail -1
