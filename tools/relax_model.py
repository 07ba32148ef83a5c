#!/usr/bin/env python3
"""CPU count model of the batched relaxation (pgq_cheapest.hip: k_relax under relax_light) on the bench's weighted
knows graph: how many vertex expansions, edge visits and label-row segments one 64-source batch costs under a given
schedule.  Counts only — no GPU, nothing the product uses; it exists to choose between schedules before GPU minutes are
spent on them.  The destination labels of every run are checked against scipy's Dijkstra.

usage: python tools/relax_model.py [--lanes 64] [--div 4] [--growth 2] [--restrict-phase-start] [--lazy M] [--band B] [--landmarks K [--prune-targets]] [--seed 6]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

INF = np.iinfo(np.int64).max // 2


def load(a):
    from duckpgq_extension_amd import graphgen
    V, s, d = graphgen.snb_knows_like(a.vertices, a.friendships, seed=100)
    w = np.random.default_rng(6).integers(1, 1000, len(s))
    off, adj, eid = graphgen.csr_from_rows(V, s, d)
    w = w[eid]
    # weight-sorted lists (ensure_weight_sorted): stable by (vertex, weight)
    owner = np.repeat(np.arange(V), np.diff(off))
    order = np.lexsort((w, owner))
    return V, off, adj[order].astype(np.int64), w[order].astype(np.int64)


def run(V, off, wadj, ws, src, dst_of, a, lb=None):
    L = len(src)
    dist = np.full((V, L), INF, dtype=np.int64)
    dirty = np.zeros((V, L), dtype=bool)
    dist[src, np.arange(L)] = 0
    dirty[src, np.arange(L)] = True
    touched = np.zeros(V, dtype=bool)
    touched[src] = True
    waited = np.zeros(V, dtype=np.int32)
    lane_ids = np.arange(L)
    w_max = int(ws.max())
    cap = max(1, int(ws.mean() / a.div))
    prev_cap = 0
    st = dict(rounds=0, phases=1, expansions=0, edge_visits=0, row_reads=0, seg_reads=0, dirty_lanes=0, improved=0)
    phase_first = False
    thr = a.band
    t0 = time.time()
    while True:
        while True:
            bound = np.array([dist[dst_of[l], l].max() if len(dst_of[l]) else 0 for l in range(L)], dtype=np.int64)
            q = np.flatnonzero(dirty.any(axis=1))
            if len(q) == 0:
                break
            live = dirty[q] & ((dist[q] if lb is None else dist[q] + lb[q]) < bound[None, :])
            if a.lazy > 0 and not phase_first:  # expand only with >= lazy dirty lanes, or after waiting 2 rounds
                few = (live.sum(axis=1) < a.lazy) & (waited[q] < 2)
                waited[q[few]] += 1
                waited[q[~few]] = 0
                hold = q[few]
            else:
                hold = np.zeros(0, dtype=np.int64)
                few = np.zeros(len(q), dtype=bool)
            nxt = np.zeros_like(dirty)
            nxt[hold] = dirty[hold]
            q, live = q[~few], live[~few]
            if a.band > 0:  # ordered rounds (relax_delta_div): labels at or above the threshold wait, the vertex stays dirty
                dvq = dist[q]
                late = live & (dvq >= thr)
                if not (live & ~late).any() and late.any():  # an empty band: straight to the smallest label left
                    thr = int(dvq[late].min()) + a.band
                    late = live & (dvq >= thr)
                nxt[q] |= late
                live = live & ~late
                thr += a.band
            keep = live.any(axis=1)
            q, live = q[keep], live[keep]
            dv = dist[q].copy()
            st["rounds"] += 1
            st["expansions"] += len(q)
            st["dirty_lanes"] += int(live.sum())
            # walk position k of every list still being walked
            pos = off[q].copy()
            if a.restrict_phase_start and phase_first:  # only the edges between the old and the new cap
                for i in range(len(q)):
                    b, e = off[q[i]], off[q[i] + 1]
                    pos[i] = b + np.searchsorted(ws[b:e], prev_cap, side="right")
            end = off[q + 1]
            act = np.arange(len(q))
            while len(act):
                inb = pos[act] < end[act]
                act = act[inb]
                if not len(act):
                    break
                k = pos[act]
                wt = ws[k]
                cand = dv[act] + wt[:, None]
                can = live[act] & (cand < bound[None, :])
                go = (wt <= cap) & can.any(axis=1)
                st["row_reads"] += len(act)  # the row of the first failing edge is read too (8 rows per trip on the GPU)
                act, k, cand, can = act[go], k[go], cand[go], can[go]
                if not len(act):
                    break
                st["edge_visits"] += len(act)
                st["seg_reads"] += int(can.reshape(len(act), L // 8 if L >= 8 else 1, -1).any(axis=2).sum())
                n = wadj[k]
                imp = can & (cand < dist[n])
                if lb is not None and a.prune_targets:  # a label that cannot lead under the bound is not written
                    imp &= (cand + lb[n]) < bound[None, :]
                r, c = np.nonzero(imp)
                if len(r):
                    np.minimum.at(dist, (n[r], c), cand[r, c])
                    nxt[n[r], c] = True
                    touched[n[r]] = True
                    st["improved"] += len(r)
                pos[act] += 1
            dirty = nxt
            phase_first = False
            if a.verbose:
                print("  round %d cap %d queue %d edges %d  %.0fs" % (st["rounds"], cap, len(q), st["edge_visits"], time.time() - t0), flush=True)
        bound = np.array([dist[dst_of[l], l].max() if len(dst_of[l]) else 0 for l in range(L)], dtype=np.int64)
        if cap >= w_max or ((bound < INF).all() and cap >= bound.max()):
            break
        prev_cap, cap = cap, cap * a.growth
        st["phases"] += 1
        phase_first = True
        tv = np.flatnonzero(touched)
        dirty[tv] = dist[tv] < INF
    st["seconds"] = round(time.time() - t0, 1)
    return dist, st


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--vertices", type=int, default=448626)
    ap.add_argument("--friendships", type=int, default=19_940_000)
    ap.add_argument("--lanes", type=int, default=64)
    ap.add_argument("--div", type=float, default=4)
    ap.add_argument("--growth", type=int, default=2)
    ap.add_argument("--restrict-phase-start", action="store_true")
    ap.add_argument("--lazy", type=int, default=0)
    ap.add_argument("--band", type=int, default=0, help="ordered rounds: width of the label band a round expands (0: off)")
    ap.add_argument("--landmarks", type=int, default=0, help="goal direction: ALT lower bounds from this many landmarks")
    ap.add_argument("--prune-targets", action="store_true", help="with --landmarks: also skip relaxations whose target cannot lead under the bound")
    ap.add_argument("--seed", type=int, default=6)
    ap.add_argument("--verbose", action="store_true")
    a = ap.parse_args()
    V, off, wadj, ws = load(a)
    pairs = np.random.default_rng(a.seed).integers(0, V, size=(a.lanes, 2))  # distinct sources: one destination per lane
    src = pairs[:, 0]
    dst_of = [np.array([pairs[l, 1]]) for l in range(a.lanes)]
    lb = None
    if a.landmarks > 0:  # ALT lower bounds, two distance arrays per landmark: lb[v, l] <= d(v, t_l)
        from scipy.sparse import csr_matrix
        from scipy.sparse.csgraph import dijkstra
        m = csr_matrix((ws.astype(np.float64), wadj, off), shape=(V, V))
        deg = np.diff(off)
        marks = np.random.default_rng(1).choice(V, size=a.landmarks, replace=False, p=deg / deg.sum())
        # the weights are per directed slot: d(v, t) >= d(v, L) - d(t, L) and >= d(L, t) - d(L, v)
        d_from = dijkstra(m, directed=True, indices=marks)                 # d(L, x)
        d_to = dijkstra(m.T.tocsr(), directed=True, indices=marks)         # d(x, L)
        lb = np.zeros((V, a.lanes), dtype=np.int64)
        t = pairs[:, 1]
        for k in range(a.landmarks):
            for x, y in ((d_to[k][:, None], d_to[k][t][None, :]), (d_from[k][t][None, :], d_from[k][:, None])):
                diff = x - y
                diff[~np.isfinite(diff)] = 0
                lb = np.maximum(lb, np.floor(diff).astype(np.int64))
    dist, st = run(V, off, wadj, ws, src, dst_of, a, lb)
    answers = [int(dist[pairs[l, 1], l]) for l in range(a.lanes)]
    print({"V": V, "E": len(wadj), **{k: v for k, v in vars(a).items() if k not in ("vertices", "friendships", "verbose")}})
    print(st)
    print("expansions per vertex %.1f, edges per expansion %.1f, dirty lanes per expansion %.1f, 64-byte segments per visited row %.2f of %d" % (
        st["expansions"] / V, st["edge_visits"] / max(st["expansions"], 1), st["dirty_lanes"] / max(st["expansions"], 1),
        st["seg_reads"] / max(st["edge_visits"], 1), max(a.lanes // 8, 1)))
    print("answers (first 8):", answers[:8], "checksum", sum(x for x in answers if x < INF))
    # every answer against scipy's Dijkstra (the generator's graph has no parallel edges, so a sparse matrix holds it)
    from scipy.sparse import csr_matrix
    from scipy.sparse.csgraph import dijkstra
    m = csr_matrix((ws.astype(np.float64), wadj, off), shape=(V, V))
    ref = dijkstra(m, directed=True, indices=src)
    want = [ref[l, pairs[l, 1]] for l in range(a.lanes)]
    ok = all((x >= INF and np.isinf(y)) or x == y for x, y in zip(answers, want))
    print("equal to scipy's Dijkstra:", ok)
    assert ok


if __name__ == "__main__":
    main()
