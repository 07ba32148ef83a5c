#!/bin/bash
# round 6, call 1: the source-centric kernel's parity tests + the cross-product bench with kernel stats
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6c1
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "ball or meet_prepass or shared_sources or golden or null_selection" > $O/pytest.txt 2>&1; tail -15 $O/pytest.txt
timeout 400 python bench.py --workload snb_cross --no-cpu-baseline --steps 5 > $O/bench_snb_cross.json 2> $O/bench_snb_cross.err; cut -c1-400 $O/bench_snb_cross.json; tail -3 $O/bench_snb_cross.err
bash tools/prof_quick.sh r6c1/prof_cross --workload snb_cross
