#!/bin/bash
# round 4, call D: the whole -m gpu suite (new SF100-scale tests included), the driver's bench command, the paths workload
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r4d
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -x -q -m gpu --durations=8 > $O/pytest_gpu.txt 2>&1; tail -14 $O/pytest_gpu.txt
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; cut -c1-400 $O/bench_default.json; tail -2 $O/bench_default.err
timeout 600 python bench.py --workload snb_paths > $O/bench_snb_paths.json 2> $O/bench_snb_paths.err; cut -c1-300 $O/bench_snb_paths.json; tail -2 $O/bench_snb_paths.err
timeout 300 python bench.py --pairs-per-gpu 8192 --no-cpu-baseline --no-legs > $O/bench_8192.json 2>/dev/null; cut -c1-200 $O/bench_8192.json
timeout 200 python tools/chunk_latency.py > $O/chunk_latency.json 2>/dev/null; cat $O/chunk_latency.json
