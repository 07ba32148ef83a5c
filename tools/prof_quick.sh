#!/bin/bash
# Quick kernel-level look on the GPU box: rocprofv3 kernel stats of a short single-stream bench run (no overlap, so the
# averages are isolated kernel durations), optionally PMC passes.  usage: tools/prof_quick.sh <tag> [pmc] [bench args]
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-q}; shift || true
PMC=0; if [ "${1:-}" = "pmc" ]; then PMC=1; shift; fi
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline $*"
PGQ_STREAMS=${PGQ_STREAMS:-1} timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- $B > $O/stats.log 2>&1
rm -f $O/stats/*kernel_trace.csv
python - <<PY
import csv,glob
for p in glob.glob("$O/stats/*kernel_stats.csv"):
    rows=list(csv.DictReader(open(p)))
    for r in rows[:14]:
        print("%-60s calls %5s avg_us %9.1f pct %5s" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
if [ $PMC = 1 ]; then
	run() { PGQ_STREAMS=1 timeout 300 rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $O/pmc_$1 -o p -- $B > $O/pmc_$1.log 2>&1; rm -f $O/pmc_$1/*kernel_trace.csv; }
	run A "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES"
	run B "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"
	run C "TCP_PENDING_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_TA_BUSY_sum SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS"
	run D "FETCH_SIZE"
	run E "WRITE_SIZE"
	run F "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR"
	python - <<PY
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(set)
for p in glob.glob("$O/pmc_*/*counter_collection.csv"):
    for r in csv.DictReader(open(p)):
        k=r["Kernel_Name"].split("(")[0].replace("void ","").replace("pgq::","")
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[(k,r["Counter_Name"])].add(r["Dispatch_Id"])
for k in agg:
    if "lanes" in k or "pull_sparse" in k or "k_push" in k or "probe" in k or "meet" in k or "ball" in k:
        print(k, {c: "%.4g"%(v/len(n[(k,c)])) for c,v in sorted(agg[k].items())})
PY
fi
