#!/bin/bash
# round 4, call E: shortestpath chain (k_meet4 on seg_walk, one wait, emission by hop), C2 and cross-leg profiles
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r4e
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden or bulk_device or full_size or fuzz or meet or unpinned or eight_shards or in_library" > $O/pytest_paths.txt 2>&1; tail -4 $O/pytest_paths.txt
for cap in 65536 16384 8192 4096; do
	PGQ_MEET_CAP_PATHS=$cap timeout 300 python bench.py --workload snb_paths --no-cpu-baseline > $O/bench_paths_$cap.json 2>/dev/null
	python - <<PY
import json
j=json.load(open("$O/bench_paths_$cap.json")); print("paths cap $cap ms", round(j["ms_per_step"],4), j["roofline_by_kernel"])
PY
done
timeout 600 python bench.py --workload rmat22 --no-cpu-baseline > $O/bench_rmat22.json 2> $O/bench_rmat22.err
python - <<PY
import json
j=json.load(open("$O/bench_rmat22.json")); print("rmat22 ms", round(j["ms_per_step"],4), j["roofline_by_kernel"], j["rows_answered_by_prepass_per_step"], j["levels_per_step"])
PY
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_rmat22 -o s -- python $R/bench.py --workload rmat22 --no-cpu-baseline --steps 10 > $O/stats_rmat22.log 2>&1; rm -f $O/stats_rmat22/*kernel_trace.csv
python - <<PY
import csv,glob
for p in glob.glob("$O/stats_rmat22/*kernel_stats.csv"):
    rows=list(csv.DictReader(open(p)))
    for r in rows[:14]: print("rmat22 %-60s calls %5s avg_us %9.1f max %9.1f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["MaxNs"])/1e3))
PY
PGQ_MEET_TRACE=1 timeout 300 python $R/bench.py --workload rmat22 --no-cpu-baseline --steps 2 --warmup 1 2>&1 >/dev/null | grep "trace" | tail -3
