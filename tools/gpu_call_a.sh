#!/bin/bash
# round 3, GPU call A: correctness of the descriptor-layout pre-pass, tuning sweeps, first bench with legs
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
O=gpurun_out/r3a; mkdir -p $O
rm -f $O/sweep.jsonl
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "meet or golden or bibfs or null_selection or fuzz or unpinned or replicas or shared_sources" > $O/pytest_meet.log 2>&1; tail -4 $O/pytest_meet.log
S="python tools/sweep_meet.py --out $O/sweep.jsonl"
timeout 300 $S --tag base --configs "meet_cap=65536;meet_cap=16384;meet_cap=4096;meet_cap=1048576;meet_align=16,meet_cap=65536;meet_align=32,meet_cap=65536" 2> $O/sweep_base.err | cut -c1-400
for v in d1 d4 m4d1 m4d4 w6; do
	PGQ_HIP_LIB=$R/build_variants/libpgq_hip_$v.so timeout 120 $S --tag $v --configs "meet_cap=65536;meet_cap=16384" 2> $O/sweep_$v.err | cut -c1-400
done
timeout 120 $S --tag p8192 --pairs 8192 --configs "meet_cap=65536" 2>/dev/null | cut -c1-400
timeout 300 $S --tag cross --cross 2048 --steps 3 --configs "trace=1;lanes=0;force_pull=2;streams=1" 2> $O/sweep_cross.err | cut -c1-600
timeout 400 python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-300 $O/bench.json; tail -3 $O/bench.err
PGQ_WBIBFS=1 timeout 200 python bench.py --workload snb_cheapest --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_cheapest_wbibfs.json 2> $O/bench_cheapest_wbibfs.err; cut -c1-200 $O/bench_cheapest_wbibfs.json
timeout 1200 python -m pytest tests -q -m gpu > $O/pytest_full.log 2>&1; tail -15 $O/pytest_full.log
