#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6c60
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "weakly or wcc or analytics or pagerank or clustering" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
PGQ_WCC_TRACE=1 timeout 600 python tools/scratch/r6_wcc.py 2>&1 | grep -i "wcc" | tail -5
