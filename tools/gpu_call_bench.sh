#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
O=gpurun_out/final; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
( time timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | grep real
python - <<'PY'
import json
d=json.load(open('gpurun_out/final/bench_default.json'))
print(d['metric'], d['value'], d['ms_per_step'], d['roofline']['frac'])
for k,v in d['legs'].items():
    print(k, round(v['ms_per_step'],3), round(v['pairs_per_s'],1), v['roofline'].get('kernel'), v['roofline'].get('frac'), (v.get('cpu_baseline') or {}).get('sample','')[:150])
PY
tail -3 $O/bench_default.err
