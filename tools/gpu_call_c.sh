#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
O=gpurun_out/r3c; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "random_graph or golden or fuzz or shared_sources or rmat18 or literal or large_inputs or unpinned" > $O/pytest_probe.log 2>&1; tail -3 $O/pytest_probe.log
PGQ_MEET=0 timeout 300 python bench.py --pairs-per-gpu 8192 --no-cpu-baseline --no-legs > $O/b_msbfs_only.json 2>/dev/null
timeout 300 python bench.py --workload snb_cross --cross-dests 32 --pairs-per-gpu 65536 --no-cpu-baseline > $O/b_cross64k.json 2>/dev/null
timeout 300 python bench.py --workload snb_cross --no-cpu-baseline > $O/b_cross2M.json 2>/dev/null
python - <<'PY'
import json
for f in ('b_msbfs_only','b_cross64k','b_cross2M'):
    d=json.load(open('gpurun_out/r3c/%s.json'%f)); print(f, round(d['ms_per_step'],4), {k:v['ms_per_step'] for k,v in d['roofline_by_kernel'].items()})
PY
