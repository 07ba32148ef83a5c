#!/bin/bash
# Round-5 measurement pass on the GPU box (via gpurun).  Part 1 (default): the driver's own command (every BASELINE config
# as a leg), the small-call / cross-product shapes, rocprofv3 kernel stats of the default workload and of the cross
# product, C-ABI call latencies.  Part 2 (PART=2): the PMC passes of both workloads.  Part 3 (PART=3): the whole -m gpu
# suite.  Part 4 (PART=4): C5 at the named scale (V = 2^28).  Everything lands under gpurun_out/pass5/; what is kept goes
# to profiles/r05/ (tools/pmc_summary.py snb_sf100 / snb_cross, tools/make_profile_readme.py r05).
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/pass5
PART=${PART:-1}
mkdir -p $O $R/gpurun_out/prof
cd $R
if [ "$PART" = 1 ]; then
T0=$(date +%s)
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "default bench rc=$? wall=$(( $(date +%s) - T0 )) s"; cut -c1-200 $O/bench_default.json
timeout 300 python bench.py --pairs-per-gpu 8192 --no-cpu-baseline --no-legs > $O/bench_snb_sf100_8192.json 2>/dev/null; cut -c1-160 $O/bench_snb_sf100_8192.json
timeout 300 python bench.py --pairs-per-gpu 2048 --no-cpu-baseline --no-legs > $O/bench_snb_sf100_2048.json 2>/dev/null; cut -c1-160 $O/bench_snb_sf100_2048.json
PGQ_MEET=0 timeout 300 python bench.py --pairs-per-gpu 8192 --no-cpu-baseline --no-legs > $O/bench_snb_sf100_8192_msbfs_only.json 2>/dev/null; cut -c1-160 $O/bench_snb_sf100_8192_msbfs_only.json
timeout 400 python bench.py --workload snb_cross --no-cpu-baseline > $O/bench_snb_cross.json 2>/dev/null; cut -c1-160 $O/bench_snb_cross.json
timeout 400 python bench.py --workload snb_cross_allv --no-cpu-baseline --steps 3 > $O/bench_snb_cross_allv.json 2>/dev/null; cut -c1-160 $O/bench_snb_cross_allv.json
timeout 400 python bench.py --workload snb_cross --cross-dests 32 --pairs-per-gpu 65536 --no-cpu-baseline > $O/bench_snb_cross_2048x32.json 2>/dev/null; cut -c1-160 $O/bench_snb_cross_2048x32.json
PGQ_MEET=0 timeout 400 python bench.py --workload snb_cross --cross-dests 32 --pairs-per-gpu 65536 --no-cpu-baseline > $O/bench_snb_cross_2048x32_lanes.json 2>/dev/null; cut -c1-160 $O/bench_snb_cross_2048x32_lanes.json
timeout 600 python bench.py --workload forest_cheapest --weights double > $O/bench_forest_cheapest_double.json 2>/dev/null; cut -c1-160 $O/bench_forest_cheapest_double.json
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_snb -o s -- python $R/bench.py --no-cpu-baseline --no-legs > $O/stats_snb.log 2>&1; rm -f $O/stats_snb/*kernel_trace.csv)
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_snb_cross -o s -- python $R/bench.py --workload snb_cross --no-cpu-baseline --steps 5 > $O/stats_snb_cross.log 2>&1; rm -f $O/stats_snb_cross/*kernel_trace.csv)
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_snb_8192 -o s -- python $R/bench.py --pairs-per-gpu 8192 --no-cpu-baseline --no-legs > $O/stats_snb_8192.log 2>&1; rm -f $O/stats_snb_8192/*kernel_trace.csv)
timeout 200 python tools/chunk_latency.py > $O/chunk_latency.json 2> $O/chunk_latency.err; cat $O/chunk_latency.json
fi
if [ "$PART" = 2 ]; then
cd /tmp && export TMPDIR=/tmp
for wl in snb_sf100 snb_cross; do
B="python $R/bench.py --workload $wl --steps 3 --warmup 0 --no-cpu-baseline --no-legs"
for pass in "B TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "D FETCH_SIZE" "E WRITE_SIZE" \
	"A SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES"; do
	set -- $pass; tag=$1; shift
	rm -rf $R/gpurun_out/prof/${wl}_$tag
	PGQ_STREAMS=1 timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/prof/${wl}_$tag -o p -- $B > $R/gpurun_out/prof/${wl}_$tag.log 2>&1
	rm -f $R/gpurun_out/prof/${wl}_$tag/*kernel_trace.csv
done
done
ls $R/gpurun_out/prof
fi
if [ "$PART" = 3 ]; then
timeout 1800 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
fi
if [ "$PART" = 4 ]; then
T0=$(date +%s)
timeout 1200 python bench.py --workload forest_cheapest --scale 28 --steps 10 > $O/bench_forest_cheapest_2_28.json 2> $O/bench_forest_cheapest_2_28.err; echo "rc=$? wall=$(( $(date +%s) - T0 )) s"; cut -c1-200 $O/bench_forest_cheapest_2_28.json
fi
ls $O
