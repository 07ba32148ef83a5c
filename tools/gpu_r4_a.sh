#!/bin/bash
# round 4, call A: the rewritten pre-pass chain — parity first (pre-pass tests, then the whole -m gpu suite), then the
# default bench, the 8192-row bench, chunk latencies and rocprofv3 kernel stats of the default workload.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r4a
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "meet or bibfs or golden or null_selection or bulk_device or unpinned" > $O/pytest_meet.txt 2>&1; tail -5 $O/pytest_meet.txt
timeout 300 python bench.py --no-legs --no-cpu-baseline > $O/bench_nolegs.json 2> $O/bench_nolegs.err; cut -c1-300 $O/bench_nolegs.json; tail -3 $O/bench_nolegs.err
timeout 300 python bench.py --pairs-per-gpu 8192 --no-cpu-baseline --no-legs > $O/bench_8192.json 2>/dev/null; cut -c1-200 $O/bench_8192.json
timeout 300 python bench.py --pairs-per-gpu 2048 --no-cpu-baseline --no-legs > $O/bench_2048.json 2>/dev/null; cut -c1-200 $O/bench_2048.json
timeout 200 python tools/chunk_latency.py > $O/chunk_latency.json 2> $O/chunk_latency.err; cat $O/chunk_latency.json
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_snb -o s -- python $R/bench.py --no-cpu-baseline --no-legs > $O/stats_snb.log 2>&1; rm -f $O/stats_snb/*kernel_trace.csv)
python - <<PY
import csv,glob
for p in glob.glob("$O/stats_snb/*kernel_stats.csv"):
    rows=list(csv.DictReader(open(p)))
    for r in rows[:10]:
        print("%-70s calls %5s avg_us %9.1f pct %5s" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
timeout 900 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -5 $O/pytest_gpu.txt
