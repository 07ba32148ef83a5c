#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6c68
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "measured_faster or first_call" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
run() { tag=$1; shift; timeout 600 python bench.py "$@" > $O/$tag.json 2> $O/$tag.err; python - <<PY
import json
d=json.loads(open("$O/$tag.json").read().strip().splitlines()[-1])
print("$tag", round(d["ms_per_step"],4), {k:round(v["ms_per_step"],4) for k,v in d["roofline_by_kernel"].items()})
PY
}
run snbx_nofc --workload snb_cross --no-cpu-baseline --no-first-call --steps 20
run snbx --workload snb_cross --no-cpu-baseline --steps 10
run rmatx_nofc --workload rmat22_cross --no-cpu-baseline --no-first-call --steps 6 --warmup 4
run rmatx --workload rmat22_cross --no-cpu-baseline --steps 6 --warmup 2
