#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6c69
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_shuffled -o s -- python $R/bench.py --workload snb_cross --cross-shuffle --no-cpu-baseline --no-first-call --steps 5 > $O/stats_shuffled.log 2>&1
rm -f $O/stats_shuffled/*kernel_trace.csv
ls $O/stats_shuffled
