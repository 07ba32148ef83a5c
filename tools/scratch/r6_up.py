import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import duckpgq_extension_amd as pgq
from duckpgq_extension_amd import graphgen
V, s, d = graphgen.snb_knows_like()
off, adj, eid = graphgen.csr_from_rows(V, s, d)
print(off.dtype, adj.dtype, eid.dtype, adj.flags["C_CONTIGUOUS"], len(adj), file=sys.stderr)
pgq.DeviceCSR(V, off, adj, eid).close()
pgq.set_option("trace", 1)
for it in range(3):
    t0 = time.perf_counter(); c = pgq.DeviceCSR(V, off, adj, eid); t1 = time.perf_counter(); c.close()
    print("host upload with ids %.2f ms" % ((t1 - t0) * 1e3), file=sys.stderr)
for it in range(3):
    t0 = time.perf_counter(); c = pgq.DeviceCSR(V, off, adj, eid, lazy_edge_ids=True); t1 = time.perf_counter(); c.close()
    print("host upload lazy ids %.2f ms" % ((t1 - t0) * 1e3), file=sys.stderr)
