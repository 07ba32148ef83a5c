#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6c25
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bibfs or bidirectional or meet or ball or rmat or far" > $O/pytest.txt 2>&1; tail -6 $O/pytest.txt
timeout 600 python bench.py --workload rmat22_cross --no-cpu-baseline --no-first-call --steps 5 > $O/bench_rmat22_cross.json 2> $O/err.txt; python - <<PY
import json
d=json.loads(open("$O/bench_rmat22_cross.json").read().strip().splitlines()[-1])
print("rmat22_cross ms", d["ms_per_step"], {k:(v["ms_per_step"]) for k,v in d["roofline_by_kernel"].items()})
PY
timeout 600 python bench.py --workload rmat22 --no-cpu-baseline --no-first-call --steps 20 > $O/bench_rmat22.json 2>> $O/err.txt; cut -c1-180 $O/bench_rmat22.json
