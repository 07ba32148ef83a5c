#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6sort
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sorted_for or source_centric or large_inputs" > $O/pytest.txt 2>&1; tail -8 $O/pytest.txt
timeout 400 python bench.py --workload snb_cross --cross-shuffle --no-cpu-baseline --steps 10 > $O/bench_shuffled.json 2> $O/bench_shuffled.err; cut -c1-400 $O/bench_shuffled.json
timeout 400 python bench.py --workload snb_cross --cross-shuffle --set ball_sort=0 --no-cpu-baseline --steps 10 > $O/bench_shuffled_nosort.json 2> $O/bench_shuffled_nosort.err; cut -c1-400 $O/bench_shuffled_nosort.json
timeout 400 python bench.py --workload snb_cross --no-cpu-baseline --steps 10 > $O/bench_grouped.json 2>/dev/null; cut -c1-300 $O/bench_grouped.json
bash tools/prof_quick.sh r6sort/prof --workload snb_cross --cross-shuffle | head -14
