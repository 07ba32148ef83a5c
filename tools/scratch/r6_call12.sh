#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6c12
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "lazy_edge or first_call or upload or bulk_device or device_csr or concurrent" > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
timeout 300 python tools/chunk_latency.py > $O/chunk_latency.json 2> $O/chunk_latency.err; cat $O/chunk_latency.json; tail -3 $O/chunk_latency.err
for t in 2 4 8 16 32; do PGQ_UPLOAD_THREADS=$t timeout 300 python - <<PY
import sys, time, os
sys.path.insert(0, "$R")
import numpy as np
import duckpgq_extension_amd as pgq
from duckpgq_extension_amd import graphgen
V, s, d = graphgen.snb_knows_like()
off, adj, eid = graphgen.csr_from_rows(V, s, d)
pgq.DeviceCSR(V, off, adj, eid).close()
r = []
for _ in range(4):
    t0 = time.perf_counter(); c = pgq.DeviceCSR(V, off, adj, eid); t1 = time.perf_counter(); c.close()
    t2 = time.perf_counter(); c = pgq.DeviceCSR(V, off, adj, None); t3 = time.perf_counter(); c.close()
    r.append(((t1 - t0) * 1e3, (t3 - t2) * 1e3))
print("threads $t: with ids %.1f ms, without %.1f ms" % (min(x[0] for x in r), min(x[1] for x in r)))
PY
done
PGQ_TRACE=1 PGQ_UPLOAD_THREADS=8 timeout 300 python tools/upload_trace.py 2>&1 | tail -25
