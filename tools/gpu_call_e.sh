#!/bin/bash
# C5 at the named scale: reply forest with V = 2^28 (SF100 message-reply has ~2.8e8 messages), 4096 pairs, int64 weights
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
O=gpurun_out/r3e; mkdir -p $O
free -g | head -2
( ulimit -v 200000000; timeout 900 python bench.py --workload forest_cheapest --scale 28 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_forest_2_28.json 2> $O/bench_forest_2_28.err ); echo rc=$?
cut -c1-700 $O/bench_forest_2_28.json; tail -3 $O/bench_forest_2_28.err
