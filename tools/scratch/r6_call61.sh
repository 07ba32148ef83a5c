#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6c61
mkdir -p $O
cd $R
PGQ_HIP_LIB=$R/build_variants/libpgq_hip_rowtrace.so timeout 600 python bench.py --workload rmat22 --no-cpu-baseline --no-first-call --steps 1 --warmup 1 > $O/t.json 2> $O/t.err
grep -h "bibfs row" $O/t.json $O/t.err | sort | uniq -c | sort -rn | head -20
grep -h "bibfs row" $O/t.json $O/t.err | wc -l
