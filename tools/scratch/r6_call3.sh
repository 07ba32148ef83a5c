#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6c3
mkdir -p $O
cd $R
PGQ_MEET_TRACE=1 timeout 400 python bench.py --workload snb_cross --no-cpu-baseline --steps 3 --warmup 1 > $O/bench_snb_cross.json 2> $O/bench_snb_cross.err; cut -c1-200 $O/bench_snb_cross.json; grep "k_src_ball trace" $O/bench_snb_cross.err | tail -3
PGQ_MEET_TRACE=1 timeout 400 python bench.py --workload snb_cross --cross-dests 32 --pairs-per-gpu 65536 --no-cpu-baseline --steps 3 --warmup 1 > $O/bench_snb_cross_2048x32.json 2>$O/x32.err; cut -c1-200 $O/bench_snb_cross_2048x32.json; grep "k_src_ball trace" $O/x32.err | tail -3
