#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
O=gpurun_out/r3c; mkdir -p $O; rm -f $O/sweep.jsonl
S="python tools/sweep_meet.py --out $O/sweep.jsonl"
P='import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d["tag"], d["cfg"], d["n"], d["wall_ms"], {k:(v["ms"],v["GBps"]) for k,v in d["kernels"].items()}, d["same_as_first"])'
timeout 200 $S --tag base --configs ";" 2>/dev/null | python -c "$P"
for v in m21 m11; do
	PGQ_HIP_LIB=$R/build_variants/libpgq_hip_$v.so timeout 120 $S --tag $v --configs ";" 2>/dev/null | python -c "$P"
done
timeout 300 $S --tag cross2M --cross 2048 --pairs 2097152 --steps 3 --configs "detect_grid_mult=16;detect_grid_mult=32;detect_grid_mult=64;detect_grid_mult=8" 2>/dev/null | python -c "$P"
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats_rmat -o s -- python $R/bench.py --workload rmat22 --no-cpu-baseline > $R/$O/stats_rmat.log 2>&1; rm -f $R/$O/stats_rmat/*kernel_trace.csv)
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r3c/stats_rmat/s_kernel_stats.csv')))
for r in rows[:14]: print("  %-70s calls %5s avg_us %9.1f pct %s" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
