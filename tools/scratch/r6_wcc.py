import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import duckpgq_extension_amd as pgq
from duckpgq_extension_amd import graphgen
V, s, d = graphgen.snb_knows_like()
off, adj, eid = graphgen.csr_from_rows(V, s, d)
for k in range(3):
    dev = pgq.DeviceCSR(V, off, adj, eid)
    t0 = time.perf_counter()
    rc = dev.L.pgq_weakly_connected_component_device(dev.h, None)
    assert rc == 0, rc
    print("wcc first call ms", (time.perf_counter() - t0) * 1e3, file=sys.stderr)
    dev.close()
