#!/bin/bash
# full GPU suite, then the general-graph cheapest_path_length timings (int64 / double, 4096 pairs, with the per-round trace of one batch)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
O=gpurun_out/final; mkdir -p $O
timeout 1100 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee $O/pytest_gpu.txt
timeout 300 python bench.py --workload snb_cheapest --weights int64 --steps 1 --warmup 1 --pairs-per-gpu 4096 > $O/bench_snb_cheapest_4096.json 2> $O/bench_snb_cheapest_4096.err; cut -c1-200 $O/bench_snb_cheapest_4096.json
timeout 300 python bench.py --workload snb_cheapest --weights double --steps 1 --warmup 1 --no-cpu-baseline --pairs-per-gpu 4096 > $O/bench_snb_cheapest_4096_double.json 2>/dev/null; cut -c1-200 $O/bench_snb_cheapest_4096_double.json
PGQ_STREAMS=1 PGQ_RELAX_TRACE=1 timeout 300 python bench.py --workload snb_cheapest --weights int64 --steps 1 --warmup 1 --no-cpu-baseline --pairs-per-gpu 64 > $O/bench_snb_cheapest_64.json 2> $O/relax_trace_64.txt; cut -c1-200 $O/bench_snb_cheapest_64.json
