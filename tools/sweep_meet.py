#!/usr/bin/env python3
"""Tuning sweep on the GPU box: the bulk iterativelength path on the bench graph under a list of option sets, one process,
one graph build.  Each line: wall ms per step (timed loop), isolated kernel-class times and algorithmic GB/s (profile
pass).  Not part of the product.

    python tools/sweep_meet.py --pairs 65536 --configs "meet_cap=65536;meet_cap=16384;meet_align=16,meet_cap=16384"
    --cross S : rows = S distinct sources x (pairs / S) destinations (the lane-batched MS-BFS path)
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import duckpgq_extension_amd as pgq  # noqa: E402
from duckpgq_extension_amd import graphgen  # noqa: E402

UPLOAD_OPTS = ("meet_align", "meet_layout", "hub_chunk", "part_weight")


def load_graph(name):
    cache = "/tmp/pgq_graph_%s.npz" % name
    if os.path.exists(cache):
        z = np.load(cache)
        return int(z["V"]), z["off"], z["adj"], z["eid"]
    if name == "snb":
        V, s, d = graphgen.snb_knows_like(448626, 19_940_000, seed=100)
    elif name.startswith("rmat"):
        V, s, d = graphgen.rmat(int(name[4:]), seed=22)
    else:
        raise SystemExit("unknown graph " + name)
    off, adj, eid = graphgen.csr_from_rows(V, s, d)
    np.savez(cache, V=V, off=off, adj=adj, eid=eid)
    return V, off, adj, eid


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graph", default="snb")
    ap.add_argument("--pairs", type=int, default=65536)
    ap.add_argument("--seed", type=int, default=4)
    ap.add_argument("--cross", type=int, default=0)
    ap.add_argument("--configs", default="")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--tag", default="")
    ap.add_argument("--out", default="gpurun_out/sweep_meet.jsonl")
    a = ap.parse_args()
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    V, off, adj, eid = load_graph(a.graph)
    rng = np.random.default_rng(a.seed)
    if a.cross:
        src = rng.choice(V, size=a.cross, replace=False)
        ps = np.repeat(src, max(1, a.pairs // a.cross))
        pd = rng.integers(0, V, len(ps))
    else:
        pr = rng.integers(0, V, size=(a.pairs, 2))
        ps, pd = pr[:, 0].copy(), pr[:, 1].copy()
    n = len(ps)
    d_src, d_dst = torch.from_numpy(ps).cuda(), torch.from_numpy(pd).cuda()
    d_out = torch.empty(n, dtype=torch.int64, device="cuda")
    t_off, t_adj = torch.from_numpy(off).cuda(), torch.from_numpy(adj).cuda()
    base = {}
    dev, dev_key, ref = None, None, None
    for spec in (a.configs.split(";") if a.configs else [""]):
        cfg = dict(kv.split("=") for kv in spec.split(",") if kv)
        for k in base:  # back to the defaults recorded at first touch
            if k not in cfg:
                pgq.set_option(k, base[k])
        pgq.load_hip().pgq_init(-1)
        for k, v in cfg.items():
            if k not in base:
                base[k] = pgq.get_option(k)
            pgq.set_option(k, v)
        key = tuple(sorted((k, v) for k, v in cfg.items() if k in UPLOAD_OPTS))
        if dev is None or key != dev_key:
            if dev is not None:
                dev.close()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            dev = pgq.DeviceCSR.from_device_ptrs(V, t_off.data_ptr(), t_adj.data_ptr())
            upload_ms = (time.perf_counter() - t0) * 1e3
            t0 = time.perf_counter()
            dev2 = pgq.DeviceCSR.from_device_ptrs(V, t_off.data_ptr(), t_adj.data_ptr())
            upload2_ms = (time.perf_counter() - t0) * 1e3
            dev2.close()
            dev_key = key

        def step():
            dev.iterativelength_bulk_ptr(n, d_src.data_ptr(), d_dst.data_ptr(), d_out.data_ptr())

        pgq.set_option("profile", 0)
        for _ in range(2):
            step()
        d_out.fill_(-7)
        pgq.reset_stats()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / a.steps * 1e3
        st = pgq.get_stats()
        res = d_out.clone()
        if ref is None:
            ref = res
        same = bool((res == ref).all())
        streams = pgq.get_option("streams")
        pgq.set_option("streams", 1)
        pgq.set_option("profile", 1)
        step()
        pgq.reset_stats()
        for _ in range(3):
            step()
        iso = pgq.get_stats()
        pgq.set_option("profile", 0)
        pgq.set_option("streams", int(streams))
        kms, kb, kl = iso["kernel_ms"], iso["algo_bytes"], iso["launches"]
        row = {"tag": a.tag, "cfg": spec, "n": n, "wall_ms": round(wall, 4), "same_as_first": same,
               "upload_ms": round(upload_ms, 2), "upload_warm_ms": round(upload2_ms, 2),
               "meet_pairs": st["meet_pairs"] / a.steps, "levels": st["levels"] / a.steps,
               "edges_scanned": st["edges_scanned"] / a.steps,
               "kernels": {k: {"ms": round(kms[k] / 3, 4), "GBps": round(kb[k] / 1e9 / (kms[k] / 1e3), 1) if kb[k] > 0 else None,
                               "launches": kl[k] / 3} for k in kms if kms[k] > 0}}
        print(json.dumps(row), flush=True)
        with open(a.out, "a") as f:
            f.write(json.dumps(row) + "\n")


if __name__ == "__main__":
    main()
