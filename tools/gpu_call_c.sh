#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
O=gpurun_out/r3c; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_full.log 2>&1; tail -6 $O/pytest_full.log
timeout 300 python bench.py --no-cpu-baseline > $O/bench.json 2>/dev/null
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3c/bench.json'))
print(d['ms_per_step'], d['roofline']['frac'], d['roofline_by_kernel'])
c=d['legs']['msbfs_cross']; print(c['ms_per_step'], c['roofline']['frac'], c['roofline']['frontier_expansion']['frac'], {k:v['ms_per_step'] for k,v in c['roofline_by_kernel'].items()})
PY
