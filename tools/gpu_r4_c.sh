#!/bin/bash
# round 4, call C: probe-first testing walk in k_meet4d, parallel estimate in k_meet_decide, chunk staging without vectors
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r4c
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "meet or bibfs or golden or null_selection or bulk_device or unpinned or full_size" > $O/pytest_meet.txt 2>&1; tail -5 $O/pytest_meet.txt
S="python tools/sweep_meet.py --steps 20 --out $O/sweep.jsonl"
show() { python - "$@" <<PY
import json,sys
for l in open("$O/sweep.jsonl"):
    r=json.loads(l)
    if r["tag"]==sys.argv[1]: print(r["tag"], r["cfg"], "n", r["n"], "wall", r["wall_ms"], "same", r["same_as_first"], "meet", r["kernels"].get("meet"), "open->levels", r["levels"])
PY
}
PGQ_MEET_TRACE=1 timeout 200 $S --tag trace --pairs 65536 --steps 3 2> $O/trace.txt >/dev/null; grep "k_meet4d trace" $O/trace.txt | tail -2
timeout 300 $S --tag b64k --pairs 65536 --configs ";meet4_grid_mult=1;meet_cap=32768;meet_grid_mult=16" > /dev/null 2>&1; show b64k
for n in 8192 2048; do
	timeout 300 $S --tag b$n --pairs $n --configs ";meet_cap_small=4096;meet_cap_small=8192;meet_small_rows=0;meet4_grid_mult=1" > /dev/null 2>&1; show b$n
	for v in ds8 ds2; do PGQ_HIP_LIB=$R/build_variants/libpgq_hip_$v.so timeout 200 $S --tag ${v}_$n --pairs $n --configs ";meet_cap_small=4096" > /dev/null 2>&1; show ${v}_$n; done
done
cd /tmp && export TMPDIR=/tmp
for n in 65536 8192 2048; do
	timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$n -o s -- python $R/tools/sweep_meet.py --steps 20 --pairs $n --out $O/sweep_prof.jsonl > $O/stats_$n.log 2>&1; rm -f $O/stats_$n/*kernel_trace.csv
	python - <<PY
import csv,glob
for p in glob.glob("$O/stats_$n/*kernel_stats.csv"):
    for r in csv.DictReader(open(p)):
        if "meet" in r["Name"] or "bibfs" in r["Name"]: print($n, "%-50s calls %5s avg_us %9.1f min %9.1f max %9.1f" % (r["Name"][:50], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
done
cd $R
timeout 200 python tools/chunk_latency.py 2>/dev/null | python -c "import json,sys; j=json.load(sys.stdin); print({k: round(v,4) for k,v in j.items() if k.startswith('iter') or k.startswith('short')})"
timeout 300 python bench.py --no-legs --no-cpu-baseline > $O/bench_nolegs.json 2> $O/bench_nolegs.err; cut -c1-300 $O/bench_nolegs.json
timeout 900 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -5 $O/pytest_gpu.txt
