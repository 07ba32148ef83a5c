#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6c20
mkdir -p $O
cd $R
bash tools/prof_quick.sh r6c20/prof_rmat --workload rmat22 --no-first-call | head -14
bash tools/prof_quick.sh r6c20/prof_8192 --workload snb_sf100 --pairs-per-gpu 8192 --no-first-call | head -12
PGQ_MEET_TRACE=1 timeout 400 python bench.py --workload rmat22 --no-cpu-baseline --no-first-call --steps 3 --warmup 1 > $O/t1.json 2> $O/t1.err; grep -i "trace" $O/t1.err | tail -4
timeout 400 python bench.py --workload rmat22 --no-cpu-baseline --no-first-call --steps 20 > $O/b_rmat.json 2>/dev/null; cut -c1-200 $O/b_rmat.json
timeout 400 python bench.py --workload snb_sf100 --pairs-per-gpu 8192 --no-cpu-baseline --no-first-call --steps 20 > $O/b_8192.json 2>/dev/null; cut -c1-200 $O/b_8192.json
