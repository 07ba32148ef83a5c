#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6c36
mkdir -p $O
cd $R
PGQ_HIP_LIB=$R/build_variants/libpgq_hip_rowtrace.so timeout 600 python bench.py --workload snb_sf100 --pairs-per-gpu 1024 --no-legs --no-cpu-baseline --no-first-call --steps 1 --warmup 1 --set meet_wide_rows_always=1 > $O/t.json 2> $O/t.err
grep -h "meet3w row" $O/t.json $O/t.err | sort -t: -k2 | head -60
grep -h "meet3w row" $O/t.json $O/t.err | wc -l
