#!/bin/bash
# Round-4 measurement pass on the GPU box (via gpurun): bench workloads, rocprofv3 kernel stats of the default bench
# command (pre-pass) and of the cross-product workload, PMC passes of the default workload for roofline.traffic, C-ABI
# call latencies, then the whole -m gpu suite.  Everything lands under gpurun_out/pass/; copy what is to be kept into
# profiles/r04/, run tools/pmc_summary.py snb_sf100 and tools/make_profile_readme.py r04.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/pass
mkdir -p $O $R/gpurun_out/prof
cd $R
timeout 600 python bench.py > $O/bench_snb_sf100.json 2> $O/bench_snb_sf100.err; cut -c1-200 $O/bench_snb_sf100.json
timeout 300 python bench.py --pairs-per-gpu 8192 --no-cpu-baseline --no-legs > $O/bench_snb_sf100_8192.json 2>/dev/null; cut -c1-160 $O/bench_snb_sf100_8192.json
timeout 300 python bench.py --pairs-per-gpu 2048 --no-cpu-baseline --no-legs > $O/bench_snb_sf100_2048.json 2>/dev/null; cut -c1-160 $O/bench_snb_sf100_2048.json
PGQ_MEET=0 timeout 300 python bench.py --pairs-per-gpu 8192 --no-cpu-baseline --no-legs > $O/bench_snb_sf100_8192_msbfs_only.json 2>/dev/null; cut -c1-160 $O/bench_snb_sf100_8192_msbfs_only.json
for wl in snb_paths forest_cheapest rmat22 snb_cross; do
	timeout 600 python bench.py --workload $wl > $O/bench_$wl.json 2> $O/bench_$wl.err; cut -c1-160 $O/bench_$wl.json
done
timeout 400 python bench.py --workload snb_cross_allv --no-cpu-baseline --steps 3 > $O/bench_snb_cross_allv.json 2>/dev/null; cut -c1-160 $O/bench_snb_cross_allv.json
timeout 400 python bench.py --workload snb_cross --cross-dests 32 --pairs-per-gpu 65536 --no-cpu-baseline > $O/bench_snb_cross_2048x32.json 2>/dev/null; cut -c1-160 $O/bench_snb_cross_2048x32.json
timeout 600 python bench.py --workload forest_cheapest --weights double > $O/bench_forest_cheapest_double.json 2>/dev/null; cut -c1-160 $O/bench_forest_cheapest_double.json
timeout 300 python bench.py --workload snb_cheapest --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_snb_cheapest_4096.json 2>/dev/null; cut -c1-160 $O/bench_snb_cheapest_4096.json
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_snb -o s -- python $R/bench.py --no-cpu-baseline --no-legs > $O/stats_snb.log 2>&1; rm -f $O/stats_snb/*kernel_trace.csv)
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_snb_cross -o s -- python $R/bench.py --workload snb_cross --no-cpu-baseline --steps 5 > $O/stats_snb_cross.log 2>&1; rm -f $O/stats_snb_cross/*kernel_trace.csv)
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_snb_8192 -o s -- python $R/bench.py --pairs-per-gpu 8192 --no-cpu-baseline --no-legs > $O/stats_snb_8192.log 2>&1; rm -f $O/stats_snb_8192/*kernel_trace.csv)
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 3 --warmup 0 --no-cpu-baseline --no-legs"
for pass in "B TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "D FETCH_SIZE" "E WRITE_SIZE" \
	"A SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" \
	"C TCP_PENDING_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_TA_BUSY_sum SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS"; do
	set -- $pass; tag=$1; shift
	rm -rf $R/gpurun_out/prof/snb_sf100_$tag
	PGQ_STREAMS=1 timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/prof/snb_sf100_$tag -o p -- $B > $R/gpurun_out/prof/snb_sf100_$tag.log 2>&1
	rm -f $R/gpurun_out/prof/snb_sf100_$tag/*kernel_trace.csv
done
cd $R
timeout 200 python tools/chunk_latency.py > $O/chunk_latency.json 2> $O/chunk_latency.err; cat $O/chunk_latency.json
timeout 1200 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
ls $O
