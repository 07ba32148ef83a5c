#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6c42
mkdir -p $O
cd $R
T0=$(date +%s)
timeout 3000 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$? wall=$(( $(date +%s) - T0 )) s"; tail -5 $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
