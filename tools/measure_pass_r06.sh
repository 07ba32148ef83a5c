#!/bin/bash
# Round-6 measurement pass on the GPU box (via gpurun).  Part 1 (default): the driver's own command (every leg), the small-call
# / cross-product shapes on their routes, rocprofv3 kernel stats, C-ABI call and upload latencies.  Part 2 (PART=2): the PMC
# passes (separate rocprofv3 --pmc runs per counter group, as MI355X_MICROARCH.md prescribes).  Everything lands under
# gpurun_out/pass6/ and gpurun_out/prof/; what is kept goes to profiles/r06/ (tools/pmc_summary.py <workload>).
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/pass6
PART=${PART:-1}
mkdir -p $O $R/gpurun_out/prof
cd $R
b() { out=$1; shift; timeout 900 env "$@" > $O/$out.json 2> $O/$out.err; cut -c1-170 $O/$out.json; }
if [ "$PART" = 1 ]; then
T0=$(date +%s)
timeout 1200 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "default bench rc=$? wall=$(( $(date +%s) - T0 )) s"; tail -c 900 $O/bench_default.json
b bench_snb_sf100_8192 python bench.py --pairs-per-gpu 8192 --no-cpu-baseline --no-legs
b bench_snb_sf100_2048 python bench.py --pairs-per-gpu 2048 --no-cpu-baseline --no-legs
b bench_snb_cross python bench.py --workload snb_cross --no-cpu-baseline
b bench_snb_cross_lanes PGQ_BALL=0 python bench.py --workload snb_cross --no-cpu-baseline
b bench_snb_cross_2048x32 python bench.py --workload snb_cross --cross-dests 32 --pairs-per-gpu 65536 --no-cpu-baseline
b bench_snb_cross_2048x128 python bench.py --workload snb_cross --cross-dests 128 --pairs-per-gpu 262144 --no-cpu-baseline
b bench_snb_cross_allv python bench.py --workload snb_cross_allv --no-cpu-baseline --steps 3
b bench_rmat22_cross python bench.py --workload rmat22_cross --no-cpu-baseline --steps 4 --warmup 2
b bench_rmat22_cross_lanes PGQ_MEET=0 python bench.py --workload rmat22_cross --no-cpu-baseline --steps 4 --warmup 2
b bench_snb_paths python bench.py --workload snb_paths --no-cpu-baseline
b bench_snb_cross_shuffled python bench.py --workload snb_cross --cross-shuffle --no-cpu-baseline
b bench_snb_cross_shuffled_nosort python bench.py --workload snb_cross --cross-shuffle --set ball_sort=0 --no-cpu-baseline
b bench_rmat22 python bench.py --workload rmat22 --no-cpu-baseline --steps 20
b bench_snb_cheapest_512 python bench.py --workload snb_cheapest --pairs-per-gpu 512 --no-cpu-baseline --steps 2 --warmup 1
b bench_snb_cheapest_512_bidir python bench.py --workload snb_cheapest --pairs-per-gpu 512 --set relax_bidir=1 --no-cpu-baseline --steps 2 --warmup 1
st() { tag=$1; shift; (cd /tmp && export TMPDIR=/tmp && timeout 400 env "$@" > $O/stats_$tag.log 2>&1; rm -f $O/stats_$tag/*kernel_trace.csv); }
st snb rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_snb -o s -- python $R/bench.py --no-cpu-baseline --no-legs --no-first-call
st snb_cross_ball rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_snb_cross_ball -o s -- python $R/bench.py --workload snb_cross --no-cpu-baseline --no-first-call --steps 5
st snb_cross PGQ_BALL=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_snb_cross -o s -- python $R/bench.py --workload snb_cross --no-cpu-baseline --no-first-call --steps 5
st snb_cross_shuffled rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_snb_cross_shuffled -o s -- python $R/bench.py --workload snb_cross --cross-shuffle --no-cpu-baseline --no-first-call --steps 5
st rmat22_cross rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_rmat22_cross -o s -- python $R/bench.py --workload rmat22_cross --no-cpu-baseline --no-first-call --steps 3 --warmup 1
timeout 300 python tools/chunk_latency.py > $O/chunk_latency.json 2> $O/chunk_latency.err; cat $O/chunk_latency.json
timeout 300 python tools/chunk_throughput.py > $O/chunk_throughput.txt 2> $O/chunk_throughput.err; cat $O/chunk_throughput.txt
fi
if [ "$PART" = 2 ]; then
cd /tmp && export TMPDIR=/tmp
for spec in "snb_sf100::--workload snb_sf100 --no-legs" "snb_cross_ball::--workload snb_cross" "snb_cross:PGQ_BALL=0:--workload snb_cross" "rmat22_cross::--workload rmat22_cross --warmup 2"; do
wl=${spec%%:*}; rest=${spec#*:}; envs=${rest%%:*}; args=${rest#*:}
B="python $R/bench.py --steps 3 --warmup 0 --no-cpu-baseline --no-first-call $args"  # (rmat22_cross: two warm-up calls — the route timing's first call and its trial — so that the profiled steps are the route it keeps)
for pass in "B TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "D FETCH_SIZE" "E WRITE_SIZE" \
	"A SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES"; do
	set -- $pass; tag=$1; shift
	rm -rf $R/gpurun_out/prof/${wl}_$tag
	env PGQ_STREAMS=1 $envs timeout 400 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/prof/${wl}_$tag -o p -- $B > $R/gpurun_out/prof/${wl}_$tag.log 2>&1
	rm -f $R/gpurun_out/prof/${wl}_$tag/*kernel_trace.csv
done
done
ls $R/gpurun_out/prof
fi
ls $O
