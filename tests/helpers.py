"""Shared helpers for the parity tests (graph feeds shaped like the reference's SQL)."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def directed_rows(edges):
    """Row feed of CreateDirectedCSRCTE (compressed_sparse_row.cpp:234-251): one row per edge-table row, in table
    order, edge id = edge rowid."""
    e = np.asarray(edges, dtype=np.int64).reshape(-1, 2)
    return e[:, 0].copy(), e[:, 1].copy(), np.arange(len(e), dtype=np.int64)


def undirected_rows(edges):
    """Row feed of CreateUndirectedCSRCTE (compressed_sparse_row.cpp:208-223): one row per distinct ordered pair in
    forward U reverse (GROUP BY src,dst), edge id = any_value -> we take the smallest contributing edge rowid, and
    emit rows sorted by (src,dst) (the reference's order is hash-aggregate order, i.e. unspecified)."""
    e = np.asarray(edges, dtype=np.int64).reshape(-1, 2)
    best = {}
    for rid, (s, d) in enumerate(e.tolist()):
        for a, b in ((s, d), (d, s)):
            if (a, b) not in best:
                best[(a, b)] = rid
    keys = sorted(best)
    src = np.array([k[0] for k in keys], dtype=np.int64)
    dst = np.array([k[1] for k in keys], dtype=np.int64)
    eid = np.array([best[k] for k in keys], dtype=np.int64)
    return src, dst, eid


def all_pairs(V):
    s, d = np.meshgrid(np.arange(V, dtype=np.int64), np.arange(V, dtype=np.int64), indexing="ij")
    return s.ravel().copy(), d.ravel().copy()


def csr_arrays_from_rows(V, src, dst):
    """offsets[V+1], adj[E], slot permutation (stable counting sort on src == reference single-thread slot order)."""
    order = np.argsort(src, kind="stable")
    off = np.zeros(V + 1, dtype=np.int64)
    np.cumsum(np.bincount(src, minlength=V), out=off[1:])
    return off, dst[order].astype(np.int64), order
