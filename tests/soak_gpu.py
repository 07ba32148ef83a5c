#!/usr/bin/env python3
"""Randomised parity soak on the GPU (test infrastructure: the oracle is the checker).  Not collected by pytest (no test_
prefix): `python tests/soak_gpu.py [seconds] [seed]` draws graphs (size, degree, skew), option sets (shipped values with the
source-centric kernel, its sort and the route timing on; caps and map placement varied) and inputs (grouped by source,
the same rows shuffled, scattered pairs, a few sources x many rows; 1 .. 300,000 rows; NULL rows) and compares
iterativelength — through the chunk API and the bulk API — with the oracle's lean restatement, row by row.  A third
argument `paths` soaks shortestpath lists and cheapest_path_length (int64 with and without zeros, double) instead.  Exit
code 1 on the first mismatch (the case is printed with its seed)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import duckpgq_extension_amd as pgq  # noqa: E402
from oracle.pgq_oracle import OracleCSR  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
import torch  # noqa: E402

t_end = time.time() + budget
case = 0
mode = sys.argv[3] if len(sys.argv) > 3 else "lengths"
while mode == "paths" and time.time() < t_end:  # shortestpath lists and cheapest_path_length (int64 and double weights)
    seed = seed0 * 100003 + case
    rng = np.random.default_rng(seed)
    case += 1
    V = int(rng.choice([300, 3000, 20000]))
    E = int(V * float(rng.choice([1.5, 5, 14])))
    if rng.random() < 0.5:
        s = (rng.random(E) ** 3 * V).astype(np.int64)
        d = (rng.random(E) ** 2 * V).astype(np.int64)
    else:
        s, d = rng.integers(0, V, E), rng.integers(0, V, E)
    e = np.arange(E, dtype=np.int64)
    wkind = int(rng.integers(0, 3))
    w = [rng.integers(1, 1000, E), rng.integers(0, 4, E), rng.random(E) + 0.01][wkind]
    opts = {"meet": int(rng.integers(0, 2)), "meet_bias": 1e9, "paths_reserve_mb": int(rng.choice([0, 1024])), "meet_cap_paths": int(rng.choice([300, 1 << 14])),
            "meet4_lds_kb": int(rng.choice([0, 150])), "relax_bidir": int(rng.integers(0, 2)), "relax_light": int(rng.choice([0, 2])),
            "relax_labels32": int(rng.integers(0, 2)), "chain": int(rng.integers(0, 2)), "streams": int(rng.choice([1, 3])),
            "relax_bidir_c0_div": int(rng.choice([1, 64, 1 << 20])), "meet_spin_wait": int(rng.integers(0, 2))}
    for k, v in opts.items():
        pgq.set_option(k, v)
    st = pgq.PgqState()
    st.build_csr(0, V, s, d, e, w)
    ora = OracleCSR.from_edges(V, s, d, e, w)
    for rep in range(2):
        n = int(rng.choice([1, 64, 700, 2500]))
        ps, pd = rng.integers(0, V, n), rng.integers(0, V, n)
        if rng.random() < 0.3:
            ps = ps[rng.integers(0, max(1, n // 50), n)]
        if st.shortestpath(0, V, ps, pd) != ora.lean_shortestpath(V, ps, pd):
            print("MISMATCH paths seed", seed, "rep", rep, "n", n, "V", V, "E", E, opts)
            sys.exit(1)
        out, ok = st.cheapest_path_length(0, V, ps, pd)
        lout, lok = ora.lean_cheapest_path_length(V, ps, pd)
        if not ((ok == lok).all() and out[ok].tobytes() == lout[ok].tobytes()):
            print("MISMATCH cheapest seed", seed, "rep", rep, "n", n, "V", V, "E", E, "weights", wkind, opts)
            sys.exit(1)
    del st
if mode == "paths":
    print("soak ok (paths + cheapest): %d graphs x 2 inputs in %.0f s" % (case, budget))
    sys.exit(0)
while time.time() < t_end:
    seed = seed0 * 100003 + case
    rng = np.random.default_rng(seed)
    case += 1
    V = int(rng.choice([300, 3000, 20000, 150000]))
    deg = float(rng.choice([1.2, 4, 12, 30]))
    E = int(V * deg)
    if rng.random() < 0.5:
        s = (rng.random(E) ** 3 * V).astype(np.int64)
        d = (rng.random(E) ** 2 * V).astype(np.int64)
    else:
        s, d = rng.integers(0, V, E), rng.integers(0, V, E)
    e = np.arange(E, dtype=np.int64)
    opts = {"meet": 1, "ball": 1, "ball_sort": 1, "route_timing": 1, "route_try_factor": float(rng.choice([0.0, 4.0])),
            "ball_seg_kb": int(rng.choice([16, 512])), "ball_cap": int(rng.choice([300, 1 << 20])), "ball_test_cap": int(rng.choice([40, 1 << 15])),
            "meet4_lds_kb": int(rng.choice([0, 150])), "ball_head_mb": int(rng.choice([0, 512])), "meet_cap": int(rng.choice([300, 1 << 14])),
            "meet_cap_small": int(rng.choice([64, 1 << 14])), "meet_wide_rows_always": int(rng.integers(0, 2)), "meet_bias": float(rng.choice([1.0, 1e9])),
            "calibration_cache": int(rng.integers(0, 2)), "bibfs_rows": int(rng.choice([0, 256])), "meet_spin_wait": int(rng.integers(0, 2))}
    for k, v in opts.items():
        pgq.set_option(k, v)
    st = pgq.PgqState()
    st.build_csr(0, V, s, d, e, None)
    ora = OracleCSR.from_edges(V, s, d, e, None)
    dev = st.device_csr(0)
    for rep in range(3):
        shape = int(rng.integers(0, 4))
        n = int(rng.choice([1, 70, 2048, 9000, 70000, 300000]))
        if shape == 0:  # scattered pairs
            ps = rng.integers(0, V, n)
        elif shape == 3:  # a few sources x many rows
            ps = np.repeat(rng.integers(0, V, max(1, n // 5000 + 1)), 5000)[:n]
        else:  # runs of random lengths (1 = grouped, 2 = the same rows shuffled)
            runs = []
            while sum(runs) < n:
                runs.append(int(rng.choice([1, 7, 300, 1024, 1500, 4000])))
            ps = np.concatenate([np.full(r, rng.integers(0, V), dtype=np.int64) for r in runs])[:n]
        n = len(ps)
        pd = rng.integers(0, V, n)
        if shape == 2:
            p = rng.permutation(n)
            ps, pd = ps[p], pd[p]
        same = rng.random(n) < 0.01
        pd[same] = ps[same]
        oln, ook = ora.lean_iterativelength(V, ps, pd, nthreads=8)
        want = np.where(ook, oln, -1)
        t_s, t_d = torch.from_numpy(ps).cuda(), torch.from_numpy(pd).cuda()
        for call in range(3):  # the second and third call run under the memos / measured routes the first one left
            t_o = torch.full((n,), -7, dtype=torch.int64, device="cuda")
            dev.iterativelength_bulk_ptr(n, t_s.data_ptr(), t_d.data_ptr(), t_o.data_ptr())
            got = t_o.cpu().numpy()
            if not (got == want).all():
                bad = np.nonzero(got != want)[0]
                print("MISMATCH bulk seed", seed, "rep", rep, "call", call, "shape", shape, "n", n, "V", V, "E", E, opts, "rows", bad[:5], got[bad[:5]], want[bad[:5]])
                sys.exit(1)
        if n <= 70000:
            ln, ok = st.iterativelength(0, V, ps, pd)
            got = np.where(ok, ln, -1)
            if not (got == want).all():
                bad = np.nonzero(got != want)[0]
                print("MISMATCH chunk seed", seed, "rep", rep, "shape", shape, "n", n, "V", V, "E", E, opts, "rows", bad[:5], got[bad[:5]], want[bad[:5]])
                sys.exit(1)
    del dev, st
print("soak ok: %d graphs x 3 inputs x 3-4 calls in %.0f s" % (case, budget))
