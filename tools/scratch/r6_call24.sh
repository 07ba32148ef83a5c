#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6c24
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "two_ended" > $O/pytest.txt 2>&1; tail -12 $O/pytest.txt
run() { tag=$1; shift; timeout 600 python bench.py --workload snb_cheapest --no-cpu-baseline --no-first-call --steps 2 --warmup 1 "$@" > $O/$tag.json 2> $O/$tag.err; python - <<PY
import json
try:
    d=json.loads(open("$O/$tag.json").read().strip().splitlines()[-1])
    print("$tag", "ms", round(d["ms_per_step"],2), "levels", d.get("levels_per_step"), "phys edges", d.get("physical_edges_scanned_per_step"))
except Exception as e:
    print("$tag failed", e); print(open("$O/$tag.err").read()[-1500:])
PY
}
run c64s128 --pairs-per-gpu 512
run c32s128 --pairs-per-gpu 512 --set relax_bidir_c0_div=32
run c16s128 --pairs-per-gpu 512 --set relax_bidir_c0_div=16
run c64s32 --pairs-per-gpu 512 --set relax_bidir_step_div=32
run full4096
PGQ_RELAX_TRACE=1 timeout 300 python bench.py --workload snb_cheapest --no-cpu-baseline --no-first-call --steps 1 --warmup 0 --pairs-per-gpu 64 --set relax_streams=1 > $O/trace.json 2> $O/trace.err; grep "^bidir" $O/trace.err | head -150 > $O/trace.txt; wc -l $O/trace.txt
