#!/usr/bin/env python3
"""CPU model of the pair-centric pre-pass's walks (k_meet3 / k_meet4d): what the choice of the expanded
endpoint, the walk cap and stage A of the distance-4 step do, counted on the bench graph without a GPU.

    python tools/walk_model.py [--rows 8192] [--vertices 448626 --friendships 19940000]

For the first `rows` bench pairs (default_rng(4)) it walks both endpoints' two-hop neighbourhoods in list order against
the other endpoint's one-hop list and reports, under the rule "expand the endpoint with the shorter one-hop LIST" (round
3) and under "... with the shorter two-hop WALK" (round 4): the mean entries walked until the first witness (in passes of
512 entries: two 1-KB requests), the rows that run into a cap, the rows proven to be at distance >= 4; and for those rows
how often the first m1 entries of the smaller two-hop neighbourhood share a vertex with the first m2 entries of the other
one (stage A of k_meet4d reads 2048 + 2048)."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def model(V, off, adj, pairs, caps=(4096, 8192, 16384, 32768), prefixes=((2048, 2048), (1024, 1024), (512, 512))):
    deg = np.diff(off)
    cs = np.concatenate([[0], np.cumsum(deg[adj])])
    work = cs[off[1:]] - cs[off[:-1]]  # entries of a vertex's two-hop walk (pgq_csr::fwork / rwork on a symmetric graph)
    mark = np.zeros(V, dtype=bool)
    rows = []
    for s, d in pairs:
        if s == d:
            continue
        Ns, Nd = adj[off[s]:off[s + 1]], adj[off[d]:off[d + 1]]
        res = {}
        for side, (A, B) in (("f", (Ns, Nd)), ("b", (Nd, Ns))):
            mark[B] = True
            seq = np.concatenate([adj[off[x]:off[x + 1]] for x in A]) if len(A) else np.zeros(0, dtype=adj.dtype)
            hit = np.flatnonzero(mark[seq])
            mark[B] = False
            res[side] = (len(seq), int(hit[0]) if len(hit) else -1)
        mark[Nd] = True
        d2 = bool(mark[Ns].any())
        mark[Nd] = False
        rows.append((s, d, len(Ns), len(Nd), res["f"][0], res["f"][1], res["b"][0], res["b"][1], int(d in Ns), int(d2)))
    r = np.array(rows, dtype=np.int64)
    s_, d_, degS, degD, wf, pf, wb, pb, d1, d2 = r.T
    rest = ~(d1 | d2).astype(bool)
    out = {"rows": len(r), "distance_le_2": float((~rest).mean())}

    def walked(tot, pos, gran=512):
        return np.where(pos >= 0, np.minimum(tot, (pos // gran + 1) * gran), tot)

    for name, fwd in (("by_list", degS <= degD), ("by_walk", wf <= wb)):
        tot, pos = np.where(fwd, wf, wb), np.where(fwd, pf, pb)
        w = walked(tot, pos)[rest]
        out[name] = {"mean_entries_walked": float(w.mean()), "distance_ge_4": int((pos[rest] < 0).sum()),
                     "cut_at_cap": {int(c): int((w > c).sum()) for c in caps}}
    # stage A on the rows proven to be at distance >= 4 under the round-4 rule
    fwd = wf <= wb
    k4 = np.flatnonzero(rest & (np.where(fwd, pf, pb) < 0))

    def prefix(v, limit):
        got, n = [], 0
        for x in adj[off[v]:off[v + 1]]:
            lst = adj[off[x]:off[x + 1]]
            got.append(lst)
            n += len(lst)
            if n >= limit:
                break
        return np.concatenate(got)[:limit] if got else np.zeros(0, dtype=adj.dtype)

    out["stage_a"] = {}
    for m1, m2 in prefixes:
        hits = 0
        for i in k4:
            a, b = (s_[i], d_[i]) if work[s_[i]] <= work[d_[i]] else (d_[i], s_[i])
            hits += len(np.intersect1d(prefix(a, m1), prefix(b, m2))) > 0
        out["stage_a"]["%dx%d" % (m1, m2)] = {"rows": int(len(k4)), "settled": int(hits)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=8192)
    ap.add_argument("--vertices", type=int, default=448626)
    ap.add_argument("--friendships", type=int, default=19_940_000)
    a = ap.parse_args()
    from duckpgq_extension_amd import graphgen
    V, s, d = graphgen.snb_knows_like(a.vertices, a.friendships, seed=100)
    off, adj, _ = graphgen.csr_from_rows(V, s, d)
    pairs = np.random.default_rng(4).integers(0, V, size=(65536, 2))[:a.rows]
    import json
    print(json.dumps(model(V, off, adj, pairs), indent=1))


if __name__ == "__main__":
    main()
