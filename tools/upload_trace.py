import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
import torch
import duckpgq_extension_amd as pgq
from duckpgq_extension_amd import graphgen
V, s, d = graphgen.snb_knows_like()
off, adj, eid = graphgen.csr_from_rows(V, s, d)
t_off, t_adj, t_eid = (torch.from_numpy(x).cuda() for x in (off, adj, eid))
pgq.set_option("trace", 1)
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    c = pgq.DeviceCSR.from_device_ptrs(V, t_off.data_ptr(), t_adj.data_ptr(), t_eid.data_ptr(), 0, 0)
    torch.cuda.synchronize(); print("upload_device ms", (time.perf_counter() - t0) * 1e3, file=sys.stderr)
    c.close()
ts, td = torch.from_numpy(s).cuda(), torch.from_numpy(d).cuda()
for it in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    c = pgq.DeviceCSR.build_from_device_rows(V, len(s), ts.data_ptr(), td.data_ptr())
    torch.cuda.synchronize(); print("build_device ms", (time.perf_counter() - t0) * 1e3, file=sys.stderr)
    c.close()
