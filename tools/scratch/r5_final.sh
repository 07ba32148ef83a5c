#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r5final
mkdir -p $O
cd $R
timeout 1800 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -4 $O/pytest_gpu.txt
T0=$(date +%s)
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "default bench rc=$? wall=$(( $(date +%s) - T0 )) s"; cut -c1-200 $O/bench_default.json
timeout 400 python bench.py --workload snb_cross --no-cpu-baseline > $O/bench_snb_cross.json 2>/dev/null; cut -c1-160 $O/bench_snb_cross.json
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_snb -o s -- python $R/bench.py --no-cpu-baseline --no-legs > $O/stats_snb.log 2>&1; rm -f $O/stats_snb/*kernel_trace.csv)
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_snb_cross -o s -- python $R/bench.py --workload snb_cross --no-cpu-baseline --steps 5 > $O/stats_snb_cross.log 2>&1; rm -f $O/stats_snb_cross/*kernel_trace.csv)
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
