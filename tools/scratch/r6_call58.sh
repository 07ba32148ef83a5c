#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6c58
mkdir -p $O
cd $R
timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline > $O/g2.json 2> $O/g2.err; echo rc=$?; tail -c 1500 $O/g2.json; tail -5 $O/g2.err
