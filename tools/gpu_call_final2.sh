#!/bin/bash
# the quick measurement pass (bench workloads + rocprofv3 kernel stats), without the 2^28 forest, plus relaxation stream counts
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/pass
mkdir -p $O
cd $R
timeout 600 python bench.py > $O/bench_snb_sf100.json 2> $O/bench_snb_sf100.err; cut -c1-200 $O/bench_snb_sf100.json
timeout 300 python bench.py --pairs-per-gpu 8192 --no-cpu-baseline --no-legs > $O/bench_snb_sf100_8192.json 2>/dev/null; cut -c1-160 $O/bench_snb_sf100_8192.json
PGQ_MEET=0 timeout 300 python bench.py --pairs-per-gpu 8192 --no-cpu-baseline --no-legs > $O/bench_snb_sf100_8192_msbfs_only.json 2>/dev/null; cut -c1-160 $O/bench_snb_sf100_8192_msbfs_only.json
for wl in snb_paths forest_cheapest rmat22 snb_cross; do
	timeout 600 python bench.py --workload $wl > $O/bench_$wl.json 2> $O/bench_$wl.err; cut -c1-160 $O/bench_$wl.json
done
timeout 400 python bench.py --workload snb_cross_allv --no-cpu-baseline --steps 3 > $O/bench_snb_cross_allv.json 2>/dev/null; cut -c1-160 $O/bench_snb_cross_allv.json
timeout 400 python bench.py --workload snb_cross --cross-dests 32 --pairs-per-gpu 65536 --no-cpu-baseline > $O/bench_snb_cross_2048x32.json 2>/dev/null; cut -c1-160 $O/bench_snb_cross_2048x32.json
timeout 600 python bench.py --workload forest_cheapest --weights double > $O/bench_forest_cheapest_double.json 2>/dev/null; cut -c1-160 $O/bench_forest_cheapest_double.json
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_snb -o s -- python $R/bench.py --no-cpu-baseline --no-legs > $O/stats_snb.log 2>&1; rm -f $O/stats_snb/*kernel_trace.csv)
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_snb_cross -o s -- python $R/bench.py --workload snb_cross --no-cpu-baseline --steps 5 > $O/stats_snb_cross.log 2>&1; rm -f $O/stats_snb_cross/*kernel_trace.csv)
for k in 6 8; do
PGQ_RELAX_STREAMS=$k timeout 300 python bench.py --workload snb_cheapest --weights int64 --steps 1 --warmup 1 --no-cpu-baseline --pairs-per-gpu 4096 > $O/bench_snb_cheapest_4096_s$k.json 2>/dev/null; cut -c1-200 $O/bench_snb_cheapest_4096_s$k.json
done
(cd /tmp && export TMPDIR=/tmp && PGQ_STREAMS=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_snb_cheapest -o s -- python $R/bench.py --workload snb_cheapest --no-cpu-baseline --steps 1 --warmup 0 --pairs-per-gpu 512 > $O/stats_snb_cheapest.log 2>&1; rm -f $O/stats_snb_cheapest/*kernel_trace.csv)
ls $O
