#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6c2
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "ball" > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
timeout 400 python bench.py --workload snb_cross --no-cpu-baseline --steps 10 > $O/bench_snb_cross.json 2> $O/bench_snb_cross.err; cut -c1-300 $O/bench_snb_cross.json; tail -3 $O/bench_snb_cross.err
timeout 400 python bench.py --workload snb_cross --cross-dests 32 --pairs-per-gpu 65536 --no-cpu-baseline > $O/bench_snb_cross_2048x32.json 2>/dev/null; cut -c1-300 $O/bench_snb_cross_2048x32.json
timeout 400 python bench.py --no-cpu-baseline --no-legs --steps 20 --warmup 5 > $O/bench_default_nolegs.json 2>/dev/null; cut -c1-300 $O/bench_default_nolegs.json
timeout 400 python bench.py --workload snb_cross_allv --no-cpu-baseline --steps 3 > $O/bench_snb_cross_allv.json 2>/dev/null; cut -c1-300 $O/bench_snb_cross_allv.json
bash tools/prof_quick.sh r6c2/prof_cross --workload snb_cross
