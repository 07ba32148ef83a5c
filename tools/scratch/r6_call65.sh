#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6c65
mkdir -p $O
cd $R
timeout 900 python tests/soak_gpu.py 120 41 paths > $O/soak_paths.txt 2>&1; echo rc=$?; tail -2 $O/soak_paths.txt | cut -c1-700
timeout 900 python tests/soak_gpu.py 300 43 > $O/soak_len.txt 2>&1; echo rc=$?; tail -2 $O/soak_len.txt | cut -c1-700
