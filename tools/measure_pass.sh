#!/bin/bash
# One measurement pass on the GPU box (via gpurun): the bench workloads, rocprofv3 kernel stats of the default bench
# command (pre-pass) and of the cross-product workload (lane-batched MS-BFS), PMC passes for roofline.traffic, chunk
# latencies, memory micro-benchmarks.  Everything lands under gpurun_out/pass/; copy what is to be kept into
# profiles/rNN/ and run tools/pmc_summary.py <workload>, tools/make_profile_readme.py rNN.
# usage: tools/measure_pass.sh [quick]
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
# configs[4] at the scale the config names (SF100 message-reply: ~2.8e8 messages): V = 2^28, every value against Dijkstra
timeout 900 python bench.py --workload forest_cheapest --scale 28 --steps 3 --warmup 1 > $O/bench_forest_cheapest_2_28.json 2> $O/bench_forest_cheapest_2_28.err; cut -c1-160 $O/bench_forest_cheapest_2_28.json
timeout 600 python bench.py --workload forest_cheapest --weights double > $O/bench_forest_cheapest_double.json 2>/dev/null; cut -c1-160 $O/bench_forest_cheapest_double.json
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_snb -o s -- python $R/bench.py --no-cpu-baseline --no-legs > $O/stats_snb.log 2>&1; rm -f $O/stats_snb/*kernel_trace.csv)
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_snb_cross -o s -- python $R/bench.py --workload snb_cross --no-cpu-baseline --steps 5 > $O/stats_snb_cross.log 2>&1; rm -f $O/stats_snb_cross/*kernel_trace.csv)
if [ "${1:-}" != "quick" ]; then
	mkdir -p $R/gpurun_out/prof
	cd /tmp && export TMPDIR=/tmp
	for cfg in "snb_sf100 --no-legs" "snb_cross --workload=snb_cross"; do
		set -- $cfg; wl=$1; extra=$2
		B="python $R/bench.py --steps 3 --warmup 0 --no-cpu-baseline $extra"
		for pass in "B TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "D FETCH_SIZE" "E WRITE_SIZE" \
			"A SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" \
			"C TCP_PENDING_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_TA_BUSY_sum SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS"; do
			set -- $pass; tag=$1; shift
			PGQ_STREAMS=1 timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/prof/${wl}_$tag -o p -- $B > $R/gpurun_out/prof/${wl}_$tag.log 2>&1
			rm -f $R/gpurun_out/prof/${wl}_$tag/*kernel_trace.csv
		done
	done
	cd $R
	# HBM ceiling of this box (copy shapes) and the FETCH_SIZE calibration for 16/32-byte gathers
	[ -x tools/membench ] || /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o tools/membench tools/membench.hip 2>/dev/null
	timeout 120 tools/membench copy > $O/membench_copy.jsonl 2>&1
	timeout 120 tools/membench segments > $O/membench_segments.jsonl 2>&1
	(cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/membench_gather -o p -- $R/tools/membench gather > $O/membench_gather.jsonl 2>&1; rm -f $O/membench_gather/*kernel_trace.csv)
	# the general-graph case of cheapest_path_length (batched relaxation, light edges first): 4096 pairs int64 / double, the
	# per-round trace of one 64-source batch, rocprofv3 kernel stats of 512 pairs with one batch in flight
	timeout 300 python bench.py --workload snb_cheapest --steps 1 --warmup 1 > $O/bench_snb_cheapest_4096.json 2>/dev/null; cut -c1-160 $O/bench_snb_cheapest_4096.json
	timeout 300 python bench.py --workload snb_cheapest --weights double --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_snb_cheapest_4096_double.json 2>/dev/null; cut -c1-160 $O/bench_snb_cheapest_4096_double.json
	PGQ_RELAX_STREAMS=1 PGQ_RELAX_TRACE=1 timeout 300 python bench.py --workload snb_cheapest --steps 1 --warmup 1 --no-cpu-baseline --pairs-per-gpu 64 > /dev/null 2> $O/relax_trace_64.txt
	(cd /tmp && export TMPDIR=/tmp && PGQ_RELAX_STREAMS=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_snb_cheapest -o s -- python $R/bench.py --workload snb_cheapest --no-cpu-baseline --steps 1 --warmup 0 --pairs-per-gpu 512 > $O/stats_snb_cheapest.log 2>&1; rm -f $O/stats_snb_cheapest/*kernel_trace.csv)
	timeout 200 python tools/chunk_latency.py > $O/chunk_latency.json 2> $O/chunk_latency.err
fi
ls $O
