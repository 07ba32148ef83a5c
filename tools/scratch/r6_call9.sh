#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6c9
mkdir -p $O
cd $R
T0=$(date +%s)
timeout 1200 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "default bench rc=$? wall=$(( $(date +%s) - T0 )) s"; tail -c 1500 $O/bench_default.json; tail -5 $O/bench_default.err
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "ball" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
