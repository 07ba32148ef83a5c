#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
O=gpurun_out/r3c; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q -k "cheapest or fuzz or weighted" 2>&1 | tail -3
PGQ_STREAMS=1 PGQ_RELAX_TRACE=1 timeout 300 python bench.py --workload snb_cheapest --weights int64 --steps 1 --warmup 1 --no-cpu-baseline --pairs-per-gpu 64 > $O/b_cheap.json 2> $O/trace_new.err
python -c "
import json; d=json.load(open('$O/b_cheap.json')); print('64', round(d['ms_per_step'],1), round(d['pairs_per_s'],1), d['roofline_by_kernel'], d['levels_per_step'], d['physical_edges_scanned_per_step'])"
for cfg in "1 int64 512 0" "3 int64 512 0" "1 int64 512 4" "1 int64 512 16" "6 int64 4096 0" "6 double 4096 0"; do
set -- $cfg
PGQ_STREAMS=$1 PGQ_RELAX_DELTA_DIV=$4 timeout 300 python bench.py --workload snb_cheapest --weights $2 --steps 1 --warmup 1 --no-cpu-baseline --pairs-per-gpu $3 > $O/b_cheap.json 2> $O/b_cheap.err
python -c "
import json; d=json.load(open('$O/b_cheap.json')); print('$3 $2 streams=$1 delta_div=$4', round(d['ms_per_step'],1), round(d['pairs_per_s'],1), d['roofline_by_kernel']['relax'], d['levels_per_step'], d['physical_edges_scanned_per_step'])"
done
