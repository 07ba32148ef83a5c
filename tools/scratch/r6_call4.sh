#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6c4
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "ball" > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
PGQ_MEET_TRACE=1 timeout 400 python bench.py --workload snb_cross --no-cpu-baseline --steps 3 --warmup 1 > $O/t1.json 2> $O/t1.err; grep "k_src_ball trace" $O/t1.err | tail -1
PGQ_MEET_TRACE=1 timeout 400 python bench.py --workload snb_cross --cross-dests 32 --pairs-per-gpu 65536 --no-cpu-baseline --steps 3 --warmup 1 > $O/t2.json 2>$O/t2.err; grep "k_src_ball trace" $O/t2.err | tail -1
timeout 400 python bench.py --workload snb_cross --no-cpu-baseline --steps 10 > $O/bench_snb_cross.json 2> $O/bench_snb_cross.err; cut -c1-250 $O/bench_snb_cross.json
timeout 400 python bench.py --workload snb_cross --cross-dests 32 --pairs-per-gpu 65536 --no-cpu-baseline > $O/bench_snb_cross_2048x32.json 2>/dev/null; cut -c1-250 $O/bench_snb_cross_2048x32.json
bash tools/prof_quick.sh r6c4/prof_cross --workload snb_cross | head -12
