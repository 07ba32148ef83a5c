#!/bin/bash
# round 4, call I: look-before-atomic on the global bit maps (R-MAT-22), sanity of the other paths
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r4i
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "meet or bibfs or c2_rmat or unpinned or large_inputs or golden" > $O/pytest_sub.txt 2>&1; tail -3 $O/pytest_sub.txt
for wl in rmat22 snb_paths; do
	timeout 400 python bench.py --workload $wl --no-cpu-baseline > $O/bench_$wl.json 2>/dev/null
	python - <<PY
import json
j=json.load(open("$O/bench_$wl.json")); print("$wl ms", round(j["ms_per_step"],4), j["roofline_by_kernel"])
PY
done
PGQ_MEET_TRACE=1 timeout 300 python $R/bench.py --workload rmat22 --no-cpu-baseline --steps 2 --warmup 1 2>&1 >/dev/null | grep "trace" | tail -1
timeout 300 python bench.py --no-legs --no-cpu-baseline > $O/bench_nolegs.json 2>/dev/null; python - <<PY
import json
j=json.load(open("$O/bench_nolegs.json")); print("default ms", round(j["ms_per_step"],4), j["roofline_by_kernel"], j["roofline"]["frac"])
PY
timeout 300 python bench.py --pairs-per-gpu 8192 --no-legs --no-cpu-baseline > $O/bench_8192.json 2>/dev/null; python - <<PY
import json
j=json.load(open("$O/bench_8192.json")); print("8192 ms", round(j["ms_per_step"],4), j["roofline_by_kernel"])
PY
