#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
O=gpurun_out/r3c; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "large_inputs or shared_sources or options_of_one or golden or eight_shards or two_ranks" > $O/pytest_sel.log 2>&1; tail -4 $O/pytest_sel.log
grep "65536 pairs" $O/pytest_sel.log
timeout 300 python bench.py --workload snb_cross --cross-dests 32 --pairs-per-gpu 65536 --no-cpu-baseline > $O/b_cross64k.json 2>/dev/null
timeout 300 python bench.py --workload snb_cross --cross-dests 128 --pairs-per-gpu 262144 --no-cpu-baseline > $O/b_cross256k.json 2>/dev/null
python - <<'PY'
import json
for f in ('b_cross64k','b_cross256k'):
    d=json.load(open('gpurun_out/r3c/%s.json'%f)); print(f, round(d['ms_per_step'],4), d['rows_answered_by_prepass_per_step'], d['levels_per_step'], {k:v['ms_per_step'] for k,v in d['roofline_by_kernel'].items()})
PY
