#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6c27
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "ball or source_centric" > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
timeout 600 python bench.py --workload rmat22_cross --no-cpu-baseline --no-first-call --steps 3 --warmup 1 --set meet_trace=1 > $O/t.json 2> $O/t.err; grep "k_src_ball trace" $O/t.err | tail -2
timeout 600 python bench.py --workload rmat22_cross --no-cpu-baseline --no-first-call --steps 5 > $O/bench_rmat22_cross.json 2> $O/err.txt; python - <<PY
import json
d=json.loads(open("$O/bench_rmat22_cross.json").read().strip().splitlines()[-1])
print("rmat22_cross ms", d["ms_per_step"], {k:(v["ms_per_step"]) for k,v in d["roofline_by_kernel"].items()})
PY
timeout 600 python bench.py --workload snb_cross --no-cpu-baseline --no-first-call --steps 10 > $O/bench_snb_cross.json 2>> $O/err.txt; cut -c1-180 $O/bench_snb_cross.json
