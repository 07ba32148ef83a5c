#!/bin/bash
# One measurement pass on the GPU box (via gpurun): parity tests, the four bench workloads, the rocprofv3 kernel
# stats of the default bench command, PMC passes for roofline.traffic, chunk latencies.  Everything lands under
# gpurun_out/pass/; copy what is to be kept into profiles/rNN/ and run tools/pmc_summary.py + make_profile_readme.py.
# usage: tools/measure_pass.sh [quick]
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/pass
mkdir -p $O
cd $R
timeout 300 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -2 $O/pytest_gpu.log
timeout 300 python bench.py > $O/bench_snb_sf100.json 2> $O/bench_snb_sf100.err; cut -c1-150 $O/bench_snb_sf100.json
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_snb -o s -- python $R/bench.py --no-cpu-baseline > $O/stats_snb.log 2>&1; rm -f $O/stats_snb/*kernel_trace.csv)
for wl in snb_paths forest_cheapest rmat22; do
	timeout 300 python bench.py --workload $wl > $O/bench_$wl.json 2> $O/bench_$wl.err; cut -c1-150 $O/bench_$wl.json
done
if [ "${1:-}" != "quick" ]; then
	mkdir -p $R/gpurun_out/prof
	cd /tmp && export TMPDIR=/tmp
	B="python $R/bench.py --workload snb_sf100 --steps 3 --warmup 0 --no-cpu-baseline"
	for pass in "B TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "D FETCH_SIZE" "E WRITE_SIZE" \
		"A SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES"; do
		set -- $pass; tag=$1; shift
		timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/prof/snb_sf100_$tag -o p -- $B > $R/gpurun_out/prof/snb_sf100_$tag.log 2>&1
		rm -f $R/gpurun_out/prof/snb_sf100_$tag/*kernel_trace.csv
	done
	cd $R
	timeout 200 python tools/chunk_latency.py > $O/chunk_latency.json 2> $O/chunk_latency.err
fi
ls $O
