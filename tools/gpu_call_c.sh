#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
O=gpurun_out/r3c; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q -k "cheapest or fuzz or weighted or c5 or concurrent or in_library" 2>&1 | tail -4
for cfg in "1 4 1" "1 4 0" "0 4 1"; do
set -- $cfg
PGQ_RELAX_TRACE=1 PGQ_RELAX_LIGHT=$1 PGQ_RELAX_LIGHT_DIV=$2 PGQ_RELAX_SPLIT=$3 timeout 300 python bench.py --workload snb_cheapest --weights int64 --steps 1 --warmup 1 --no-cpu-baseline --pairs-per-gpu 64 > $O/b_cheap.json 2> $O/trace_$1_$2_$3.err
python -c "
import json; d=json.load(open('$O/b_cheap.json')); print('64 light=$1 div=$2 split=$3', round(d['ms_per_step'],1), round(d['pairs_per_s'],1), d['roofline_by_kernel'], d['levels_per_step'], d['physical_edges_scanned_per_step'])"
PGQ_RELAX_LIGHT=$1 PGQ_RELAX_LIGHT_DIV=$2 PGQ_RELAX_SPLIT=$3 timeout 300 python bench.py --workload snb_cheapest --weights int64 --steps 1 --warmup 0 --no-cpu-baseline --pairs-per-gpu 512 > $O/b_cheap.json 2> $O/b_cheap.err
python -c "
import json; d=json.load(open('$O/b_cheap.json')); print('512 light=$1 div=$2 split=$3', round(d['ms_per_step'],1), round(d['pairs_per_s'],1), d['roofline_by_kernel'], d['levels_per_step'], d['physical_edges_scanned_per_step'])"
done
