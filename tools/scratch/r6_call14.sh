#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6c14
mkdir -p $O
cd $R
T0=$(date +%s)
timeout 2400 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$? wall=$(( $(date +%s) - T0 )) s"; tail -5 $O/pytest_gpu.txt
T0=$(date +%s)
timeout 1200 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "default bench rc=$? wall=$(( $(date +%s) - T0 )) s"; tail -c 1300 $O/bench_default.json; tail -3 $O/bench_default.err
