#!/bin/bash
# round-5 call A: parity of the rewritten level loop / lane assignment / detection, then the cross leg
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r5a
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "random_graph_all_variants or cross_product or golden or fuzz or prepass or defer or traversed or probe or unpinned or lanes or hub or null" > $O/pytest_a.txt 2>&1; tail -5 $O/pytest_a.txt
timeout 300 python bench.py --workload snb_cross --no-cpu-baseline --steps 10 > $O/bench_snb_cross.json 2> $O/bench_snb_cross.err; cut -c1-300 $O/bench_snb_cross.json
PGQ_SPEC_LEVELS=0 timeout 300 python bench.py --workload snb_cross --no-cpu-baseline --steps 10 > $O/bench_snb_cross_nospec.json 2>/dev/null; cut -c1-200 $O/bench_snb_cross_nospec.json
PGQ_SORT_SINGLE_BATCH=1 PGQ_SPEC_LEVELS=0 timeout 300 python bench.py --workload snb_cross --no-cpu-baseline --steps 10 > $O/bench_snb_cross_sorted_nospec.json 2>/dev/null; cut -c1-200 $O/bench_snb_cross_sorted_nospec.json
PGQ_TRACE=1 timeout 300 python bench.py --workload snb_cross --no-cpu-baseline --steps 2 --warmup 1 > /dev/null 2> $O/trace_cross.txt; tail -30 $O/trace_cross.txt
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_snb_cross -o s -- python $R/bench.py --workload snb_cross --no-cpu-baseline --steps 5 > $O/stats_snb_cross.log 2>&1; rm -f $O/stats_snb_cross/*kernel_trace.csv)
ls $O $O/stats_snb_cross
PGQ_MAX_WORDS=16 timeout 300 python bench.py --workload snb_cross --no-cpu-baseline --steps 10 > $O/bench_snb_cross_w16.json 2>/dev/null; cut -c1-200 $O/bench_snb_cross_w16.json
