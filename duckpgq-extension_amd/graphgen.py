"""Synthetic graphs shaped like BASELINE.json's configs — numpy, deterministic by seed.

All generators return (V, src, dst) edge-table rows in "table order"; `csr_from_rows` then reproduces the
reference's single-threaded CSR (stable counting sort on src == arrival order per vertex,
csr_creation.cpp:132-138).  Generated arrays are cached under $PGQ_CACHE (default /tmp/pgq_cache).
"""
import hashlib
import os

import numpy as np

_CACHE = os.environ.get("PGQ_CACHE", "/tmp/pgq_cache")


def _cached(name, fn):
    path = os.path.join(_CACHE, name + ".npz")
    if os.path.exists(path):
        try:
            z = np.load(path)
            return tuple(z[k] for k in z.files)
        except Exception:
            pass
    out = fn()
    try:
        os.makedirs(_CACHE, exist_ok=True)
        np.savez(path, *out)
    except Exception:
        pass
    return out


def rmat(scale, edge_factor=16, a=0.57, b=0.19, c=0.19, d=0.05, seed=22):
    """Graph500-style R-MAT, directed, duplicates and self-loops kept, no vertex permutation (C2)."""
    del d

    def gen():
        rng = np.random.default_rng(seed)
        n_edges = edge_factor << scale
        src = np.zeros(n_edges, dtype=np.int64)
        dst = np.zeros(n_edges, dtype=np.int64)
        ab, abc = a + b, a + b + c
        for bit in range(scale):
            r = rng.random(n_edges, dtype=np.float32)
            src |= (r >= ab).astype(np.int64) << bit
            dst |= (((r >= a) & (r < ab)) | (r >= abc)).astype(np.int64) << bit
        return src, dst

    src, dst = _cached("rmat_s%d_e%d_seed%d" % (scale, edge_factor, seed), gen)
    return 1 << scale, src, dst


def snb_knows_like(V=448626, friendships=19_940_000, seed=100, gamma=2.6, max_degree=4000):
    """LDBC-SNB-SF100-shaped Person-knows-Person: Chung-Lu graph with a truncated power-law expected degree,
    mean degree ~2*friendships/V, symmetrised + de-duplicated, rows sorted by (src, dst) (C3/C4)."""

    def gen():
        rng = np.random.default_rng(seed)
        u = rng.random(V)
        wgt = (1.0 - u) ** (-1.0 / (gamma - 1.0))  # Pareto tail
        wgt = np.minimum(wgt, max_degree / (2.0 * friendships / V) * 1.0)
        cdf = np.cumsum(wgt)
        cdf /= cdf[-1]
        m = int(friendships * 1.06)  # a few percent are lost to duplicates / self loops
        a_ = np.searchsorted(cdf, rng.random(m)).astype(np.int64)
        b_ = np.searchsorted(cdf, rng.random(m)).astype(np.int64)
        keep = a_ != b_
        a_, b_ = a_[keep], b_[keep]
        lo, hi = np.minimum(a_, b_), np.maximum(a_, b_)
        key = np.unique(lo * V + hi)[:friendships]
        lo, hi = key // V, key % V
        s = np.concatenate([lo, hi])
        t = np.concatenate([hi, lo])
        order = np.argsort(s * V + t, kind="stable")
        return s[order], t[order]

    src, dst = _cached("snbknows_V%d_F%d_seed%d" % (V, friendships, seed), gen)
    return V, src, dst


def reply_forest(V=1 << 24, root_fraction=0.2, seed=5):
    """SNB-shaped Message-replyOf-Message forest: every non-root picks a uniformly random earlier vertex as its
    parent; directed child -> parent like the reference's edge table (C5)."""

    def gen():
        rng = np.random.default_rng(seed)
        is_root = rng.random(V) < root_fraction
        is_root[0] = True
        child = np.nonzero(~is_root)[0].astype(np.int64)
        parent = (rng.random(len(child)) * child).astype(np.int64)
        return child, parent

    src, dst = _cached("replyforest_V%d_seed%d" % (V, seed), gen)
    return V, src, dst


def csr_from_rows(V, src, dst):
    """offsets[V+1], adj[E], edge_ids[E] with the reference's single-thread slot order."""
    src = np.asarray(src, dtype=np.int64)
    order = np.argsort(src, kind="stable")
    off = np.zeros(V + 1, dtype=np.int64)
    np.cumsum(np.bincount(src, minlength=V), out=off[1:])
    return off, np.asarray(dst, dtype=np.int64)[order], order.astype(np.int64)


def graph_fingerprint(off, adj):
    h = hashlib.sha256()
    h.update(np.ascontiguousarray(off).tobytes())
    h.update(np.ascontiguousarray(adj).tobytes())
    return h.hexdigest()[:16]


def degree_summary(off):
    deg = np.diff(off)
    return {"V": int(len(off) - 1), "E": int(off[-1]), "mean": float(deg.mean()) if len(deg) else 0.0,
            "max": int(deg.max()) if len(deg) else 0, "zero": int((deg == 0).sum()),
            "p50": float(np.percentile(deg, 50)) if len(deg) else 0.0,
            "p99": float(np.percentile(deg, 99)) if len(deg) else 0.0}
