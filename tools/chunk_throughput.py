#!/usr/bin/env python3
"""What a DuckDB query gets through the scalar-function API: T native worker threads, each calling pgq_iterativelength on its
own 2048-row DataChunk (host vectors in and out, one C-ABI call per chunk, as the expression executor does —
iterativelength.cpp:34) over one shared device CSR.  The threads are native (tools/chunk_mt.cpp, built here with g++): Python
threads re-take the GIL after every 70-us call and measure the GIL.  This script writes the SF100-shaped CSR to /tmp and runs it."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from duckpgq_extension_amd import graphgen  # noqa: E402

exe = os.path.join(ROOT, "tools", "chunk_mt")
csrc = os.path.join(ROOT, "duckpgq-extension_amd", "csrc")
if not os.path.exists(exe):
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-o", exe, os.path.join(ROOT, "tools", "chunk_mt.cpp"),
                           "-L" + csrc, "-lpgq_hip", "-Wl,-rpath," + csrc, "-pthread"])
V, s, d = graphgen.snb_knows_like()
off, adj, eid = graphgen.csr_from_rows(V, s, d)
off.tofile("/tmp/pgq_off.bin")
adj.tofile("/tmp/pgq_adj.bin")
sys.stdout.write(subprocess.check_output([exe, "/tmp/pgq_off.bin", "/tmp/pgq_adj.bin"] + sys.argv[1:], env=dict(os.environ, LD_LIBRARY_PATH=csrc)).decode())
