#!/bin/bash
# round 4, last call: device-side small rounds under relax_light — weighted tests first, then what the driver runs at round end
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r4final2
mkdir -p $O
cd $R
timeout 400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "cheapest or weighted or c5" > $O/pytest_cheapest.txt 2>&1; tail -2 $O/pytest_cheapest.txt
python - > $O/ring.txt 2>&1 <<PY
# a weighted ring with chords: hundreds of relaxation rounds with a handful of changed vertices each
import time, numpy as np
import duckpgq_extension_amd as pgq
from oracle.pgq_oracle import OracleCSR
V = 20000
rng = np.random.default_rng(3)
s = np.concatenate([np.arange(V), np.arange(V), rng.integers(0, V, 200)])
d = np.concatenate([(np.arange(V) + 1) % V, (np.arange(V) + 7) % V, rng.integers(0, V, 200)])
order = np.argsort(s, kind="stable"); s, d = s[order], d[order]
off = np.concatenate([[0], np.cumsum(np.bincount(s, minlength=V))]).astype(np.int64)
w = rng.integers(1, 50, len(s))
ps, pd = rng.integers(0, V, 64), rng.integers(0, V, 64)
ora = OracleCSR.adopt(V, off, d.astype(np.int64), np.arange(len(s), dtype=np.int64), w)
want, wok = ora.lean_cheapest_path_length(V, ps, pd)
for light in (1, 0):
    pgq.set_option("relax_light", light)
    dev = pgq.DeviceCSR(V, off, d.astype(np.int64), np.arange(len(s), dtype=np.int64), w)
    dev.cheapest_path_length(ps, pd)
    pgq.reset_stats()
    t0 = time.perf_counter(); out, ok = dev.cheapest_path_length(ps, pd); dt = time.perf_counter() - t0
    st = pgq.get_stats()
    print("relax_light", light, "ms", round(dt * 1e3, 2), "rounds", st["levels"], "equal", bool((ok == wok).all() and (out[ok] == want[wok]).all()))
    dev.close()
PY
cat $O/ring.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1200 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -2 $O/pytest_gpu.txt
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; cut -c1-200 $O/bench_default.json
python - <<PY
import json
j=json.load(open("$O/bench_default.json")); l=j["legs"]["cheapest_general"]; print("cheapest_general ms", l["ms_per_step"], l["cpu_baseline"]["sample"][-60:])
PY
