#!/bin/bash
# round 4, last call: what the driver runs at round end, on the final tree — smoke(), the -m gpu suite, the default bench line
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r4final
mkdir -p $O
cd $R
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 1200 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; cut -c1-330 $O/bench_default.json
