#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6c53
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "measured_faster" > $O/pytest.txt 2>&1; grep -n "AssertionError\|seen\|passed\|failed" $O/pytest.txt | tail -8
