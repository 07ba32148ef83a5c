#!/bin/bash
# Runs on the GPU box (via gpurun): separate rocprofv3 --pmc passes over one bench step, CSVs under gpurun_out/prof/.
# usage: tools/collect_pmc.sh <workload> [extra bench args]
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
WL=${1:-snb_sf100}; shift || true
mkdir -p $R/gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --workload $WL --steps 3 --warmup 0 --no-cpu-baseline $*"
run() { timeout 400 rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $R/gpurun_out/prof/${WL}_$1 -o p -- $B > $R/gpurun_out/prof/${WL}_$1.log 2>&1; rm -f $R/gpurun_out/prof/${WL}_$1/*kernel_trace.csv; }
run A "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES"
run B "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"
run C "TCP_PENDING_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_TA_BUSY_sum SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS"
run D "FETCH_SIZE"
run E "WRITE_SIZE"
