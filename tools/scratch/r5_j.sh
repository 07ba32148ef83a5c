#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r5j
mkdir -p $O
cd $R
timeout 1800 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -4 $O/pytest_gpu.txt
T0=$(date +%s)
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "default bench rc=$? wall=$(( $(date +%s) - T0 )) s"; cut -c1-200 $O/bench_default.json
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
