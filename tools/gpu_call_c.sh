#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
O=gpurun_out/r3c; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "cheapest or fuzz or weighted or c5" > $O/pytest_cheap.log 2>&1; tail -3 $O/pytest_cheap.log
for div in 64 16 256 0; do
for wt in int64 double; do
PGQ_RELAX_DELTA_DIV=$div timeout 200 python bench.py --workload snb_cheapest --weights $wt --steps 1 --warmup 0 --no-cpu-baseline --pairs-per-gpu 512 > $O/b_cheap_${wt}_$div.json 2> $O/b_cheap_${wt}_$div.err
python -c "
import json; d=json.load(open('$O/b_cheap_${wt}_$div.json')); print('$wt div=$div', d['ms_per_step'], d['pairs_per_s'], d['roofline_by_kernel'], d['levels_per_step'], d['physical_edges_scanned_per_step'])"
done; done
