#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6c63
mkdir -p $O
cd $R
timeout 900 python tests/soak_gpu.py 300 5 paths > $O/soak.txt 2>&1; echo rc=$?; tail -5 $O/soak.txt | cut -c1-600
