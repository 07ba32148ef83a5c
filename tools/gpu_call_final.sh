#!/bin/bash
# full GPU suite, then the cross-product and default bench lines
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
O=gpurun_out/final; mkdir -p $O
timeout 1100 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee $O/pytest_gpu.txt
timeout 300 python bench.py --workload snb_cross --no-cpu-baseline > $O/bench_snb_cross.json 2>/dev/null; cut -c1-200 $O/bench_snb_cross.json
