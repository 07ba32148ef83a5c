#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6c32
mkdir -p $O
cd $R
PGQ_HIP_LIB=$R/build_variants/libpgq_hip_rowtrace.so timeout 600 python bench.py --workload rmat22 --no-cpu-baseline --no-first-call --steps 1 --warmup 1 > $O/t.json 2> $O/t.err
grep -h "meet3 row" $O/t.json $O/t.err | sort | uniq -c | sort -rn | head -12
run() { wl=$1; tag=$2; shift; shift; timeout 600 python bench.py --workload $wl --no-cpu-baseline --no-first-call --steps 20 "$@" > $O/$tag.json 2> $O/$tag.err; python - <<PY
import json
try:
    d=json.loads(open("$O/$tag.json").read().strip().splitlines()[-1])
    print("$tag", "ms", round(d["ms_per_step"],4), {k:round(v["ms_per_step"],4) for k,v in d["roofline_by_kernel"].items()})
except Exception as e:
    print("$tag failed", e); print(open("$O/$tag.err").read()[-800:])
PY
}
run rmat22 rmat
run snb_sf100 snb --no-legs
run rmat22_cross rmatx --steps 5
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "meet or prepass or golden or rmat or path" > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
