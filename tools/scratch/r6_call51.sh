#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6c51
mkdir -p $O
cd $R
run() { tag=$1; shift; timeout 600 env "$@" > $O/$tag.json 2> $O/$tag.err; python - <<PY
import json
try:
    d=json.loads(open("$O/$tag.json").read().strip().splitlines()[-1])
    print("$tag", "ms", round(d["ms_per_step"],4), "levels", d["levels_per_step"], {k:round(v["ms_per_step"],3) for k,v in d["roofline_by_kernel"].items()}, "frac", round(d["roofline"]["frac"],3), round(d["roofline"]["step"]["frac"],3), d.get("first_call",{}).get("first_call_ms_all"))
except Exception as e:
    print("$tag failed", e); print(open("$O/$tag.err").read()[-800:])
PY
}
run rmatx X=1 python bench.py --workload rmat22_cross --no-cpu-baseline --steps 6 --trace-steps
grep "rmat22_cross step" $O/rmatx.err | cut -c1-200
run rmatx_lanes PGQ_MEET=0 python bench.py --workload rmat22_cross --no-cpu-baseline --no-first-call --steps 4 --warmup 2
run snbx X=1 python bench.py --workload snb_cross --no-cpu-baseline --steps 10
run snb X=1 python bench.py --no-legs --no-cpu-baseline --steps 10
run snbx128 X=1 python bench.py --workload snb_cross --cross-dests 128 --pairs-per-gpu 262144 --no-cpu-baseline --no-first-call
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "ball or source_centric or sorted_for or probe or first_call" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
