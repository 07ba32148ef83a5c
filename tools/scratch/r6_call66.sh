#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6c66
mkdir -p $O
cd $R
timeout 600 python bench.py --workload snb_cross --no-cpu-baseline --no-first-call --steps 3 --warmup 1 --set meet_trace=1 > $O/t.json 2> $O/t.err; grep "k_src_ball trace" $O/t.err | tail -2 | cut -c1-600
