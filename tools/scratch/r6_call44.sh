#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6c44
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "ball or source_centric or sorted_for" > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
run() { wl=$1; tag=$2; shift; shift; timeout 600 python bench.py --workload $wl --no-cpu-baseline --no-first-call --steps 6 "$@" > $O/$tag.json 2> $O/$tag.err; python - <<PY
import json
try:
    d=json.loads(open("$O/$tag.json").read().strip().splitlines()[-1])
    print("$tag", "ms", round(d["ms_per_step"],4), {k:round(v["ms_per_step"],4) for k,v in d["roofline_by_kernel"].items()})
except Exception as e:
    print("$tag failed", e); print(open("$O/$tag.err").read()[-800:])
PY
}
run rmat22_cross rmatx
run rmat22_cross rmatx_trace --set meet_trace=1 --steps 2
grep "k_src_ball trace" $O/rmatx_trace.err | tail -2
run snb_cross snbx --steps 10
