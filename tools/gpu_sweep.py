#!/usr/bin/env python3
"""Tuning sweep run on the GPU box: times the bulk iterativelength path across lane-word widths / direction
thresholds on the bench graphs and writes a table under gpurun_out/.  Not part of the product."""
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


def load_graph(name):
    if name == "snb":
        V, s, d = graphgen.snb_knows_like()
    elif name.startswith("rmat"):
        V, s, d = graphgen.rmat(int(name[4:]), seed=22)
    else:
        raise SystemExit("unknown graph " + name)
    off, adj, eid = graphgen.csr_from_rows(V, s, d)
    return V, off, adj, eid


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graph", default="snb")
    ap.add_argument("--pairs", type=int, default=8192)
    ap.add_argument("--words", default="4,8,16")
    ap.add_argument("--push_div", default="24")
    ap.add_argument("--modes", default="0")
    ap.add_argument("--bpc", default="8")
    ap.add_argument("--defer", default="8")
    ap.add_argument("--force_pull", default="0")
    ap.add_argument("--sparse_unroll", default="2")
    ap.add_argument("--sparse_lds", default="1")
    ap.add_argument("--sparse_pw", default="1")
    ap.add_argument("--extra", default="", help="key=value[,key=value] options set once")
    ap.add_argument("--streams", default="2")
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--trace", type=int, default=0)
    ap.add_argument("--out", default="gpurun_out/sweep.jsonl")
    a = ap.parse_args()
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    t0 = time.time()
    V, off, adj, eid = load_graph(a.graph)
    print("graph %s V=%d E=%d gen %.1fs" % (a.graph, V, off[-1], time.time() - t0), flush=True)
    dev = pgq.DeviceCSR(V, off, adj, None)
    rng = np.random.default_rng(4)
    pairs = rng.integers(0, V, (a.pairs, 2))
    d_src = torch.from_numpy(pairs[:, 0].copy()).cuda()
    d_dst = torch.from_numpy(pairs[:, 1].copy()).cuda()
    d_out = torch.empty(a.pairs, dtype=torch.int64, device="cuda")
    d_te = torch.empty(a.pairs, dtype=torch.int64, device="cuda")
    dev.traversed_edges_bulk_ptr(a.pairs, d_src.data_ptr(), d_dst.data_ptr(), d_out.data_ptr(), d_te.data_ptr())
    te = int(d_te.sum().item())
    ref = d_out.cpu().numpy().copy()
    print("TE=%d reachable=%d mean_len=%.2f" % (te, int((ref >= 0).sum()), ref[ref > 0].mean()), flush=True)
    print("copy bw GB/s:", pgq.copy_bandwidth_gbps(1 << 30, 5), flush=True)
    pgq.set_option("profile", 1)
    for kv in [x for x in a.extra.split(",") if x]:
        k, v = kv.split("=")
        pgq.set_option(k, float(v) if "." in v else int(v))
    pgq.set_option("trace", a.trace)
    with open(a.out, "a") as f:
        for mode in [int(x) for x in a.modes.split(",")]:
            for bpc, nstr in [(int(x), int(y)) for x in a.bpc.split(",") for y in a.streams.split(",")]:
                pgq.set_option("streams", nstr)
                for words in [int(x) for x in a.words.split(",")]:
                  for dfr in [int(x) for x in a.defer.split(",")]:
                   for fp in [int(x) for x in a.force_pull.split(",")]:
                    for pd_, su, sv in [(float(x), int(y), int(z)) for x in a.push_div.split(",") for y in a.sparse_unroll.split(",") for z in a.sparse_pw.split(",")]:
                        pgq.set_option("sparse_unroll", su)
                        pgq.set_option("sparse_pw", sv)
                        pgq.set_option("sparse_lds", int(a.sparse_lds))
                        pgq.set_option("defer", dfr)
                        pgq.set_option("force_pull", fp)
                        pgq.set_option("words", words)
                        pgq.set_option("push_div", pd_)
                        pgq.set_option("force_mode", mode)
                        pgq.set_option("blocks_per_cu", bpc)
                        best = None
                        for _ in range(a.reps):
                            pgq.reset_stats()
                            torch.cuda.synchronize()
                            t = time.perf_counter()
                            dev.iterativelength_bulk_ptr(a.pairs, d_src.data_ptr(), d_dst.data_ptr(), d_out.data_ptr())
                            torch.cuda.synchronize()
                            dt = time.perf_counter() - t
                            if best is None or dt < best[0]:
                                best = (dt, pgq.get_stats())
                        okk = bool((d_out.cpu().numpy() == ref).all())
                        dt, st = best
                        row = {"graph": a.graph, "pairs": a.pairs, "words": words, "push_div": pd_, "mode": mode, "defer": dfr, "force_pull": fp, "sparse_unroll": su, "sparse_pw": sv, "extra": a.extra,
                               "bpc": bpc, "streams": nstr, "ms": dt * 1e3, "mteps": te / dt / 1e6, "pairs_per_s": a.pairs / dt,
                               "match": okk, "levels": st["levels"], "push": st["push_levels"],
                               "pull": st["pull_levels"], "kernel_ms": st["kernel_ms"],
                               "algo_gb": {k: v / 1e9 for k, v in st["algo_bytes"].items() if v},
                               "edges_scanned": st["edges_scanned"]}
                        kms = st["kernel_ms"]
                        for k in ("pull", "push", "pull_sparse"):
                            if kms.get(k, 0) > 0:
                                row[k + "_GBps"] = st["algo_bytes"][k] / 1e9 / (kms[k] / 1e3)
                        f.write(json.dumps(row) + "\n")
                        f.flush()
                        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
