#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r5e
mkdir -p $O
cd $R
T0=$(date +%s)
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
echo "rc=$? wall=$(( $(date +%s) - T0 )) s"
tail -5 $O/bench_default.err; cut -c1-300 $O/bench_default.json
