#!/bin/bash
# round 3: full GPU test-suite, the bench workloads, rocprofv3 kernel stats + PMC passes of the default bench command
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
O=gpurun_out/r3d; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_full.log 2>&1; tail -8 $O/pytest_full.log
timeout 600 python bench.py > $O/bench_snb_sf100.json 2> $O/bench_snb_sf100.err; cut -c1-250 $O/bench_snb_sf100.json; tail -2 $O/bench_snb_sf100.err
for wl in snb_paths forest_cheapest rmat22; do
	timeout 400 python bench.py --workload $wl > $O/bench_$wl.json 2> $O/bench_$wl.err; cut -c1-200 $O/bench_$wl.json
done
timeout 300 python bench.py --pairs-per-gpu 8192 --no-cpu-baseline --no-legs > $O/bench_snb_sf100_8192.json 2>/dev/null; cut -c1-200 $O/bench_snb_sf100_8192.json
(cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats_snb -o s -- python $R/bench.py --no-cpu-baseline > $R/$O/stats_snb.log 2>&1; rm -f $R/$O/stats_snb/*kernel_trace.csv)
head -12 $O/stats_snb/*kernel_stats.csv | cut -c1-150
