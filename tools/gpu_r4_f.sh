#!/bin/bash
# round 4, call F: seg_walk with requests in flight across rounds
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r4h
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden or bulk_device or full_size or fuzz or meet or unpinned or bibfs or c2_rmat or clustering or eight_shards or large_inputs" > $O/pytest_sub.txt 2>&1; tail -4 $O/pytest_sub.txt
S="python tools/sweep_meet.py --steps 20 --out $O/sweep.jsonl"
for n in 65536 8192 2048; do timeout 200 $S --tag b$n --pairs $n > /dev/null 2>&1; done
python - <<PY
import json
for l in open("$O/sweep.jsonl"):
    r=json.loads(l); print(r["tag"], "wall", r["wall_ms"], "same", r["same_as_first"], {k:v for k,v in r["kernels"].items() if k in ("meet","meet4","bibfs")})
PY
for wl in snb_paths rmat22; do
	timeout 400 python bench.py --workload $wl --no-cpu-baseline > $O/bench_$wl.json 2>/dev/null
	python - <<PY
import json
j=json.load(open("$O/bench_$wl.json")); print("$wl ms", round(j["ms_per_step"],4), j["roofline_by_kernel"], "upload_ms", j["config"]["csr_upload_ms"])
PY
done
PGQ_MEET_TRACE=1 timeout 300 python $R/bench.py --workload rmat22 --no-cpu-baseline --steps 2 --warmup 1 2>&1 >/dev/null | grep "trace" | tail -2
timeout 300 python bench.py --no-legs --no-cpu-baseline > $O/bench_nolegs.json 2>/dev/null; python - <<PY
import json
j=json.load(open("$O/bench_nolegs.json")); print("default ms", round(j["ms_per_step"],4), j["roofline_by_kernel"], j["roofline"]["frac"], j["roofline"]["prepass_chain"])
PY
