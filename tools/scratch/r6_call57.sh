#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6c57
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "written_again or path or golden" > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
timeout 300 python bench.py --workload snb_paths --no-cpu-baseline --no-first-call > $O/b.json 2>/dev/null; cut -c1-200 $O/b.json
