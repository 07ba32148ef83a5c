"""CPU models of device schedules whose exactness rests on an argument rather than on the reference's loop order:
k_bibfs (pgq_meet.hip: level-synchronous bidirectional BFS per row), k_wbibfs (pgq_cheapest.hip: bidirectional
band-wise label correcting per row), the packed list walk of the pre-pass (pgq_walk.h) and k_relax under relax_light
(pgq_cheapest.hip: rounds over weight-sorted lists under a doubling cap with per-lane bounds).  The models follow the kernels' control flow (side selection, termination tests,
far-queue refill with band skipping, duplicates in the near queue, arbitrary relaxation order) and are compared with the
oracle's per-pair BFS / Dijkstra.  They exist to pin the *schedule*; the kernels themselves are checked on the GPU
(tests/test_gpu_parity.py: test_bibfs_few_open_rows_any_distance, test_weighted_pair_search_bit_exact)."""
import numpy as np

from oracle.pgq_oracle import OracleCSR

INF = 1 << 62


def _csr(V, s, d, w=None):
    order = np.argsort(s, kind="stable")
    adj = d[order]
    off = np.zeros(V + 1, dtype=np.int64)
    np.add.at(off, s[order] + 1, 1)
    off = np.cumsum(off)
    ro = np.argsort(adj, kind="stable")  # the upload's permutation: in-edges by (source, slot)
    radj = s[order][ro]
    roff = np.zeros(V + 1, dtype=np.int64)
    np.add.at(roff, adj[ro] + 1, 1)
    roff = np.cumsum(roff)
    ww = w[order] if w is not None else None
    return off, adj, roff, radj, ww, (ww[ro] if w is not None else None), order


def bibfs_model(off, adj, roff, radj, s, d):
    """k_bibfs: expand the side whose frontier has fewer adjacency entries; first meeting = a + b + 1."""
    if s == d:
        return 0
    seen = [{s}, {d}]
    front = [[s], [d]]
    lvl = [0, 0]
    X = [(off, adj), (roff, radj)]
    while True:
        work = [sum(int(X[k][0][v + 1] - X[k][0][v]) for v in front[k]) for k in (0, 1)]
        side = 0 if work[0] <= work[1] else 1
        o, a = X[side]
        nxt, hit = [], False
        for v in front[side]:
            for u in a[o[v]:o[v + 1]].tolist():
                if u in seen[side ^ 1]:
                    hit = True
                if u not in seen[side]:
                    seen[side].add(u)
                    nxt.append(u)
        if hit:
            return lvl[0] + lvl[1] + 1
        if not nxt:
            return None  # this side's closure is complete
        front[side] = nxt
        lvl[side] += 1


def wbibfs_model(off, adj, w, roff, radj, rw, s, d, delta, rng, prune=False):
    """k_wbibfs: bands of width delta, near queue relaxed to its fixpoint, far queue takes a vertex once."""
    if s == d:
        return 0
    dist = [{s: 0}, {d: 0}]
    near, far, r, best = [[s], [d]], [[], []], [0, 0], INF
    X = [(off, adj, w), (roff, radj, rw)]
    while True:
        side = 0 if r[0] <= r[1] else 1
        o, a, ww = X[side]
        rn = r[side] + delta
        while near[side]:
            cur = near[side]
            rng.shuffle(cur)
            nxt = []
            for v in cur:
                dv = dist[side][v]
                if prune and dv + r[side ^ 1] >= best:  # wbibfs_prune: cannot start a better path
                    continue
                for e in range(int(o[v]), int(o[v + 1])):
                    u, nd = int(a[e]), dv + int(ww[e])
                    old = dist[side].get(u, INF)
                    if nd < old:
                        dist[side][u] = nd
                        other = dist[side ^ 1].get(u, INF)
                        if other != INF:
                            best = min(best, nd + other)
                        if nd < rn:
                            nxt.append(u)
                        elif old == INF:  # first labelling beyond the band: the only time a vertex enters the far queue
                            far[side].append(u)
            near[side] = nxt
        r[side] = rn
        if r[0] + r[1] >= best:
            return best
        live = [u for u in far[side] if dist[side][u] >= r[side]]
        if not live:
            return best if best != INF else None
        m = min(dist[side][u] for u in live)
        if m >= r[side] + delta:
            r[side] = (m // delta) * delta
            if r[0] + r[1] >= best:
                return best
        lo = r[side]
        near[side] = [u for u in live if dist[side][u] < lo + delta]
        far[side] = [u for u in live if dist[side][u] >= lo + delta]


def test_bidirectional_bfs_schedule_matches_oracle():
    rng = np.random.default_rng(7)
    for trial in range(60):
        V = int(rng.integers(4, 80))
        E = int(rng.integers(V // 2, V * 4))
        s, d = rng.integers(0, V, E), rng.integers(0, V, E)
        off, adj, roff, radj, _, _, order = _csr(V, s, d)
        ora = OracleCSR.adopt(V, off, adj, np.arange(E, dtype=np.int64))
        ps, pd = rng.integers(0, V, 40), rng.integers(0, V, 40)
        ln, ok = ora.lean_iterativelength(V, ps, pd)
        for a, b, want, k in zip(ps.tolist(), pd.tolist(), ln.tolist(), ok.tolist()):
            assert bibfs_model(off, adj, roff, radj, a, b) == (want if k else None)


def test_bidirectional_band_search_schedule_matches_dijkstra():
    rng = np.random.default_rng(11)
    for trial in range(60):
        V = int(rng.integers(5, 60))
        E = int(rng.integers(V, V * 6))
        s, d = rng.integers(0, V, E), rng.integers(0, V, E)
        w = rng.integers(0 if trial % 3 == 0 else 1, 40, E)  # every third graph has zero-weight edges
        off, adj, roff, radj, ww, rw, order = _csr(V, s, d, w)
        ora = OracleCSR.adopt(V, off, adj, np.arange(E, dtype=np.int64), ww.astype(np.int64))
        ps, pd = rng.integers(0, V, 30), rng.integers(0, V, 30)
        out, ok = ora.lean_cheapest_path_length(V, ps, pd)
        for a, b, want, k in zip(ps.tolist(), pd.tolist(), out.tolist(), ok.tolist()):
            delta = int(rng.integers(1, 30))
            assert wbibfs_model(off, adj, ww, roff, radj, rw, a, b, delta, rng) == (want if k else None)
            assert wbibfs_model(off, adj, ww, roff, radj, rw, a, b, delta, rng, prune=True) == (want if k else None)


def prepass_path_model(off, adj, roff, radj, s, d):
    """Inner vertices the path variants of the pre-pass pick (k_meet3<true>, k_meet4<true>): distances 1..4 only.
    Backward walks over in-lists ordered by source; the first hit in walk order is taken."""
    outs = set(adj[off[s]:off[s + 1]].tolist())
    if s == d:
        return [s]
    if d in outs:
        return [s, d]
    ind = radj[roff[d]:roff[d + 1]].tolist()  # ascending by source
    for y in ind:  # distance 2: smallest common neighbour
        if y in outs:
            return [s, y, d]
    for y in ind:  # distance 3: first in-list (ascending y) with a hit, first hit in it (ascending x)
        for x in radj[roff[y]:roff[y + 1]].tolist():
            if x in outs:
                return [s, x, y, d]
    two = set()
    for x in outs:
        two.update(adj[off[x]:off[x + 1]].tolist())
    marked = outs | two  # k_meet4: B = N_out(src) + N_out(N_out(src))
    for y in ind:  # distance 4: third vertex y, second vertex x; first vertex = smallest in-neighbour of x src points at
        for x in radj[roff[y]:roff[y + 1]].tolist():
            if x in marked:
                v1 = next(z for z in radj[roff[x]:roff[x + 1]].tolist() if z in outs)
                return [s, v1, x, y, d]
    return None


def test_ordered_backward_walk_picks_the_reference_path():
    # the claim behind the early exit of the path kernels: with in-lists ordered by source, the first witness a backward
    # walk meets is the path the reference's min-id-parent rule reconstructs (shortest_path.cpp:21-31)
    rng = np.random.default_rng(23)
    checked = {2: 0, 3: 0, 4: 0}
    for trial in range(40):
        V = int(rng.integers(8, 70))
        E = int(rng.integers(V, V * 5))
        s, d = rng.integers(0, V, E), rng.integers(0, V, E)
        off, adj, roff, radj, _, _, order = _csr(V, s, d)
        eid = np.arange(E, dtype=np.int64)
        ora = OracleCSR.adopt(V, off, adj, eid)
        ps, pd = rng.integers(0, V, 60), rng.integers(0, V, 60)
        for a, b, path in zip(ps.tolist(), pd.tolist(), ora.lean_shortestpath(V, ps, pd)):
            if path is None or len(path) > 9:
                continue
            got = prepass_path_model(off, adj, roff, radj, a, b)
            assert got == path[0::2]  # the vertices of [v, e, v, ...]
            checked[(len(path) - 1) // 2] = checked.get((len(path) - 1) // 2, 0) + 1
    assert checked[2] and checked[3] and checked[4]


# ---- padded adjacency + packed walk (pgq_runtime.hip build_meet_layout, pgq_walk.h seg_walk) -------------------------------

def padded_layout_model(off, adj, align):
    """k_seg_groups / k_seg_fill / k_fill_padded / k_fill_desc: per vertex {first group, entries}, lists padded to whole
    16-byte groups (rounded up to `align` entries) with copies of their last entry, one descriptor per adjacency slot."""
    V = len(off) - 1
    ln = np.diff(off)
    ng = ((ln + align - 1) // align) * (align // 4)
    gbeg = np.concatenate([[0], np.cumsum(ng)])
    padj = np.full(int(gbeg[-1]) * 4, -7, dtype=np.int64)
    for v in range(V):
        for g in range(int(gbeg[v]), int(gbeg[v + 1])):  # one thread per group, clamped to the last entry
            i0 = (g - int(gbeg[v])) * 4
            for k in range(4):
                padj[4 * g + k] = adj[off[v] + min(i0 + k, ln[v] - 1)]
    desc = [(int(u), int(gbeg[u]), int(ln[u])) for u in adj.tolist()]
    return gbeg[:-1], ln, padj, desc


def dpp_incl_scan_model(x, op):
    """wave_incl_scan_u32 / wave_incl_max_u32 (pgq_walk.h): row_shr 1,2,3 of the input, row_shr 4 / 8 of the running
    value under bank masks, row_bcast 15 into rows 1 and 3, row_bcast 31 into rows 2 and 3; 0 for out-of-row reads."""
    x = list(x)

    def shr(v, k, banks=None):  # lane i reads lane i - k of its row of 16; disabled banks read the identity 0
        out = []
        for i in range(64):
            ok = (i % 16) >= k and (banks is None or ((i % 16) // 4) in banks)
            out.append(v[i - k] if ok else 0)
        return out

    r = [op(a, b) for a, b in zip(x, shr(x, 1))]
    r = [op(a, b) for a, b in zip(r, shr(x, 2))]
    r = [op(a, b) for a, b in zip(r, shr(x, 3))]
    r = [op(a, b) for a, b in zip(r, shr(r, 4, banks=(1, 2, 3)))]
    r = [op(a, b) for a, b in zip(r, shr(r, 8, banks=(2, 3)))]
    r = [op(r[i], r[(i // 16) * 16 - 1]) if (i // 16) in (1, 3) else r[i] for i in range(64)]
    r = [op(r[i], r[31]) if (i // 16) in (2, 3) else r[i] for i in range(64)]
    return r


def seg_owner_model(P, ng, x0):
    """seg_owner: every non-empty list overlapping [x0, x0 + 64) writes lane + 1 at the window position where it begins
    (0 if it began earlier); an inclusive max-scan spreads the ids."""
    win = [0] * 64
    for lane in range(64):
        start = int(P[lane]) - ng[lane]
        if ng[lane] and start < x0 + 64 and int(P[lane]) > x0:
            pos = start - x0 if start > x0 else 0
            assert win[pos] == 0
            win[pos] = lane + 1
    return [v - 1 for v in dpp_incl_scan_model(win, max)]


def seg_walk_model(desc_list, padj, w, nw, depth, stop_after=None):
    """seg_walk: rounds of 64 descriptors (lane j = descriptor j of the round); wavefront w of nw takes the requests
    w, w + nw, ... of every round; virtual groups 64 c + lane, owners by the window scan, lanes past the end re-reading
    the last group (ok = False).  Returns the (entry, ev, ok) stream and the number of groups requested."""
    seen, groups = [], 0
    n = len(desc_list)
    pb = 0
    while pb < n:
        d = [desc_list[pb + l] if pb + l < n else (0, 0, 0) for l in range(64)]
        ng = [(x[2] + 3) >> 2 for x in d]
        P = np.cumsum(ng)
        assert dpp_incl_scan_model(ng, lambda a, b: a + b) == [int(v) for v in P]
        total = int(P[63])
        D = [d[l][1] - (int(P[l]) - ng[l]) for l in range(64)]
        nchunk = (total + 63) >> 6
        nxt, issued, halted = w, 0, False
        while nxt < nchunk:
            for _ in range(depth):
                if nxt >= nchunk:
                    break
                owner = seg_owner_model(P, ng, nxt * 64)
                for lane in range(64):
                    xx = nxt * 64 + lane
                    ok = xx < total
                    xs = xx if ok else total - 1
                    j = owner[lane]
                    assert (int(P[j]) - ng[j]) <= xs < int(P[j])  # the window scan found the list holding group xs
                    g = D[j] + xs
                    for k in range(4):
                        seen.append((int(padj[4 * g + k]), d[j][0], ok))
                nxt += nw
                issued += 1
            if stop_after is not None and len(seen) >= stop_after:
                halted = True
                break
        g = issued * 64  # 64 groups per request, except the round's last one
        if issued and nxt - nw == nchunk - 1:
            g -= nchunk * 64 - total
        groups += g
        if halted:
            break
        pb += 64
    return seen, groups


def test_padded_layout_and_packed_walk_visit_exactly_the_lists():
    rng = np.random.default_rng(11)
    for align in (4, 8, 16, 32):
        V = 90
        deg = rng.integers(0, 12, V)
        deg[rng.integers(0, V, 10)] = 0
        deg[3] = 150  # a list longer than two requests
        off = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
        adj = rng.integers(0, V, int(off[-1]))
        gbeg, ln, padj, desc = padded_layout_model(off, adj, align)
        assert (padj >= 0).all()
        for v in range(V):  # every padded list = the list, then copies of its last entry, start aligned
            assert (gbeg[v] * 4) % align == 0
            got = padj[4 * gbeg[v]:4 * gbeg[v] + ((ln[v] + 3) // 4) * 4]
            assert got[:ln[v]].tolist() == adj[off[v]:off[v + 1]].tolist()
            assert all(x == adj[off[v + 1] - 1] for x in got[ln[v]:])
        for v in (3, 5, 17):  # two-hop walk of v: one wavefront, and 16 wavefronts with stride 16
            lst = desc[off[v]:off[v + 1]]
            want = sorted((int(x), u) for (u, _, _) in lst for x in adj[off[u]:off[u + 1]].tolist())
            for nw in (1, 16):
                got, total = [], 0
                for w in range(nw):
                    seen, g = seg_walk_model(lst, padj, w, nw, depth=2)
                    got += [(x, ev) for x, ev, ok in seen if ok]
                    total += g
                # the wavefronts' requests partition the rounds' groups
                assert total == sum((ln_u + 3) // 4 for (_, _, ln_u) in lst)
                # padding repeats real entries of the same list: the set of (entry, expanded vertex) pairs is exact,
                # and every real entry is visited at least once
                assert set(got) == set(want)
                import collections
                cg, cw = collections.Counter(got), collections.Counter(want)
                assert all(cg[k] >= cw[k] for k in cw)
            seen, g = seg_walk_model(lst, padj, 0, 1, depth=2, stop_after=1)  # early exit after the first pass
            assert (0 < g <= 128) if want else g == 0


# ---- round 4: requests in flight across rounds, the filter + register set, the two-ended queue (pgq_walk.h, pgq_meet.hip) ----

def seg_walk_v3_model(desc_list, w, nw, depth, stop_after_requests=None, max_entries=None):
    """seg_walk since round 4: DEPTH request slots that outlive a round of 64 descriptors.  A pass processes every busy
    slot (refilling it from the CURRENT round first); the top-up at the head of the loop is the one place a round ends and
    the next begins, and fills slots left empty from the following rounds.  Returns the (round start, request) pairs in
    the order they were PROCESSED, the order they were ISSUED, and `resume` (first descriptor of the earliest round with
    an unprocessed request when the walk was cut)."""
    n = len(desc_list)
    rounds = []
    for pb in range(0, n, 64):
        total = sum((ln + 3) >> 2 for (_, _, ln) in desc_list[pb:pb + 64])
        rounds.append((pb, (total + 63) >> 6, total))
    ri, nxt = 0, w
    slots = [None] * depth
    issued, processed, requested = [], [], 0
    is_open = n > 0
    resume = 0

    def issue(u):
        nonlocal nxt, requested
        pb, nchunk, total = rounds[ri]
        slots[u] = (pb, nxt)
        issued.append((pb, nxt))
        requested += 4 * min(64, total - nxt * 64)
        nxt += nw

    halt = False
    while True:
        while is_open:  # top-up
            if all(x is not None for x in slots):
                break
            if nxt >= rounds[ri][1]:
                ri += 1
                if ri >= len(rounds):
                    is_open = False
                    break
                nxt = w
                continue
            for u in range(depth):
                if slots[u] is None and nxt < rounds[ri][1]:
                    issue(u)
        any_chunk = False
        for u in range(depth):
            if slots[u] is None:
                continue
            any_chunk = True
            cur = slots[u]
            slots[u] = None
            if is_open and nxt < rounds[ri][1]:
                issue(u)
            processed.append(cur)
        if not any_chunk:
            break
        if stop_after_requests is not None and len(processed) >= stop_after_requests:
            halt = True
            break
        if max_entries is not None and requested > max_entries:
            halt = True
            break
    if halt:
        resume = rounds[ri][0] if is_open else (rounds[-1][0] if rounds else 0)
        for x in slots:
            if x is not None:
                resume = min(resume, x[0])
    return processed, issued, resume


def test_requests_in_flight_across_rounds_cover_every_group_once_and_in_order():
    rng = np.random.default_rng(12)
    for trial in range(40):
        n = int(rng.integers(1, 400))
        lens = rng.integers(0, 9, n)  # many short lists: rounds of one or two requests, the case the overlap is for
        if trial % 3 == 0:
            lens[rng.integers(0, n, 3)] = rng.integers(200, 700, 3)
        desc = [(i, 0, int(l)) for i, l in enumerate(lens)]
        want = []
        for pb in range(0, n, 64):
            total = sum((l + 3) >> 2 for (_, _, l) in desc[pb:pb + 64])
            want += [(pb, c) for c in range((total + 63) >> 6)]
        for nw, depth in ((1, 2), (1, 4), (16, 2)):
            got = []
            for w in range(nw):
                processed, issued, _ = seg_walk_v3_model(desc, w, nw, depth)
                assert processed == sorted(processed) and issued == sorted(issued)  # walk order, per wavefront
                assert sorted(processed) == sorted(issued)
                got += processed
            assert sorted(got) == want  # every request of every round exactly once
        # a cut walk: everything before `resume` has been processed, so the bit-map kernel may take the walk up there
        processed, issued, resume = seg_walk_v3_model(desc, 0, 1, 2, max_entries=int(rng.integers(1, 4000)))
        done = set(processed)
        assert all(r in done for r in want if r[0] < resume)
        assert resume % 64 == 0 and 0 <= resume <= max(0, (n - 1) // 64 * 64)


def filter_word(x):
    return (x >> 5) & 1023


def filter_mask(x, bigv):
    b = ((x >> 15) ^ (x >> 20) ^ (x >> 25)) if bigv else (x >> 15)
    return (1 << (x & 31)) | (1 << (b & 31))


def test_two_bit_filter_has_no_false_negatives_and_few_false_positives():
    """flt_word / flt_mask / flt_test (pgq_walk.h): both bits of an id sit in one of 1024 words; a set member always
    passes (the exact test in registers then decides), and for ids below 2^20 only cross-combinations of two members
    that share a word can pass wrongly."""
    rng = np.random.default_rng(13)
    for bigv, vmax in ((False, 448626), (True, 1 << 28)):
        for size in (1, 100, 512):
            members = rng.choice(vmax, size=size, replace=False)
            bm = [0] * 1024
            for x in members.tolist():
                bm[filter_word(x)] |= filter_mask(x, bigv)
            test = lambda x: (bm[filter_word(x)] & filter_mask(x, bigv)) == filter_mask(x, bigv)
            assert all(test(x) for x in members.tolist())
            probes = rng.integers(0, vmax, 200000)
            mset = set(members.tolist())
            fp = sum(1 for x in probes.tolist() if x not in mset and test(x))
            assert fp / len(probes) < (0.002 if not bigv else 0.004) * max(1, size / 100) ** 2


def test_two_ended_queue_positions_and_dynamic_hand_out():
    """queue_push / queue_pos / k_meet4d's job counter (pgq_meet.hip): long rows from the front, proven-distance-4 rows
    from the back of one array; jobs are handed out in ascending order = every long row before any other."""
    rng = np.random.default_rng(14)
    cap = 1000
    kinds = rng.random(700) < 0.05  # True: a long row
    front = back = 0
    arr = [None] * cap
    for i, long_row in enumerate(kinds.tolist()):
        if long_row:
            arr[front] = ("long", i)
            front += 1
        else:
            arr[cap - 1 - back] = ("known4", i)
            back += 1
    n = front + back
    pos = lambda j: j if j < front else cap - 1 - (j - front)
    order = [arr[pos(j)] for j in range(n)]
    assert all(x is not None for x in order) and len({x[1] for x in order}) == n
    assert [k for k, _ in order] == ["long"] * front + ["known4"] * back
    # a grid of G workgroups: the first G positions are the workgroups' own, the rest drawn from a counter
    G = 64
    drawn = list(range(G)) + [G + t for t in range(n)]  # counter values may run past n: those workgroups stop
    assert sorted(j for j in drawn if j < n) == list(range(n))


def light_first_model(V, off, adj, w, lanes, dests, cap0, rng, dtype):
    """k_relax under relax_light (pgq_cheapest.hip): lists sorted by weight, Jacobi rounds over the prefix under a cap
    that doubles per phase; a lane expands a vertex only while its label is under the lane's bound (largest tentative
    label among its destinations, refreshed per round); a vertex's walk stops at the first edge above the cap or whose
    candidate gets no dirty lane under its bound; a phase ends at its fixpoint, the search when the cap has reached the
    largest weight or every lane's bound.  Vertices of a round are taken in random order and read the labels as they are
    (the kernel's wavefronts run concurrently over one label array)."""
    inf = np.iinfo(np.int64).max if dtype == np.int64 else np.inf
    L = len(lanes)
    wsorted, wadj = w.copy(), adj.copy()
    for v in range(V):
        o = np.argsort(w[off[v]:off[v + 1]], kind="stable")
        wsorted[off[v]:off[v + 1]] = w[off[v]:off[v + 1]][o]
        wadj[off[v]:off[v + 1]] = adj[off[v]:off[v + 1]][o]
    dist = np.full((V, L), inf, dtype=dtype)
    dirty = np.zeros((V, L), dtype=bool)
    for l, s in enumerate(lanes):
        dist[s, l] = 0
        dirty[s, l] = True
    touched = set(int(s) for s in lanes)
    w_max = wsorted.max() if len(wsorted) else 0
    cap = dtype(cap0)
    while True:
        while dirty.any():  # rounds of a phase
            bound = np.array([max((dist[d, l] for d in dests[l]), default=dtype(0)) for l in range(L)], dtype=dtype)
            queue = np.flatnonzero(dirty.any(axis=1))
            rng.shuffle(queue)
            nxt = np.zeros_like(dirty)
            for v in queue.tolist():
                mine = dirty[v] & (dist[v] < bound)
                dv = dist[v].copy()
                if not mine.any():
                    continue
                for k in range(off[v], off[v + 1]):
                    wt = wsorted[k]
                    cand = dv + wt if dtype != np.int64 else np.where(dv == inf, inf, dv + np.where(dv == inf, 0, wt))
                    if wt > cap or not (mine & (cand < bound)).any():
                        break
                    n = wadj[k]
                    imp = mine & (cand < bound) & (cand < dist[n])
                    dist[n, imp] = cand[imp]
                    nxt[n] |= imp
                    if imp.any():
                        touched.add(int(n))
            dirty = nxt
        bound = np.array([max((dist[d, l] for d in dests[l]), default=dtype(0)) for l in range(L)], dtype=dtype)
        if cap >= w_max or ((bound < inf).all() and cap >= bound.max()):
            break
        cap = cap + cap
        for v in touched:  # k_redirty: every labelled vertex again, over the longer prefix
            dirty[v] = dist[v] < inf
    return dist


def test_light_edges_first_schedule_matches_dijkstra():
    rng = np.random.default_rng(23)
    checked = 0
    for trial in range(50):
        V = int(rng.integers(5, 50))
        E = int(rng.integers(V, V * 7))
        s, d = rng.integers(0, V, E), rng.integers(0, V, E)
        dtype = np.int64 if trial % 2 == 0 else np.float64
        if dtype == np.int64:
            w = rng.integers(0 if trial % 4 == 0 else 1, 60, E).astype(np.int64)  # zero weights every other int graph
        else:
            w = rng.random(E) * (10.0 ** rng.integers(-3, 4, E))  # sums that round: the fold order matters
        off, adj, _, _, ww, _, order = _csr(V, s, d, w)
        ora = OracleCSR.adopt(V, off, adj, np.arange(E, dtype=np.int64), ww)
        L = int(rng.integers(1, 7))
        lanes = rng.choice(V, size=min(L, V), replace=False)
        dests = [rng.integers(0, V, int(rng.integers(0, 4))).tolist() for _ in lanes]  # a lane may have no destination left
        cap0 = max(1, int(w.mean() / 4)) if dtype == np.int64 else max(float(w.mean()) / 4, 1e-300)
        dist = light_first_model(V, off, adj, ww, lanes, dests, cap0, rng, dtype)
        for l, src in enumerate(lanes.tolist()):
            if not dests[l]:
                continue
            ps = np.full(len(dests[l]), src, dtype=np.int64)
            out, ok = ora.lean_cheapest_path_length(V, ps, np.asarray(dests[l], dtype=np.int64))
            for dd, want, k in zip(dests[l], out.tolist(), ok.tolist()):
                got = dist[dd, l]
                if not k:
                    assert got == (np.iinfo(np.int64).max if dtype == np.int64 else np.inf)
                elif dtype == np.int64:
                    assert int(got) == want
                else:
                    assert np.float64(got).tobytes() == np.float64(want).tobytes()  # bit for bit
                checked += 1
    assert checked > 150


def test_relax_count_model_tool_runs_and_agrees_with_dijkstra():
    """tools/relax_model.py (the count model DESIGN 7 quotes) on a small graph: every variant ends with the answers of
    scipy's Dijkstra (the tool asserts it) — restricted phase start, ordered bands, landmark bounds."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for extra in ([], ["--restrict-phase-start"], ["--band", "8"], ["--landmarks", "4", "--prune-targets"], ["--lazy", "4"]):
        out = subprocess.run([sys.executable, os.path.join(root, "tools", "relax_model.py"), "--vertices", "3000",
                              "--friendships", "40000", "--lanes", "16"] + extra, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        assert "equal to scipy's Dijkstra: True" in out.stdout


def test_walk_model_tool_runs_and_the_walk_size_rule_is_never_worse_on_average():
    """tools/walk_model.py (the CPU model DESIGN.md 3.0 quotes for the expansion-side rule and for stage A of the
    distance-4 step) on a small SNB-shaped graph: it runs, both rules agree on which rows are within two hops, and
    expanding the endpoint with the shorter two-hop walk never walks more on average than the shorter-list rule."""
    import importlib.util
    import os
    from duckpgq_extension_amd import graphgen
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("walk_model", os.path.join(root, "tools", "walk_model.py"))
    wm = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(wm)
    V, s, d = graphgen.snb_knows_like(6000, 90000, seed=5)
    off, adj, _ = graphgen.csr_from_rows(V, s, d)
    pairs = np.random.default_rng(1).integers(0, V, size=(400, 2))
    out = wm.model(V, off, adj, pairs, caps=(256, 1024), prefixes=((512, 512),))
    assert out["rows"] > 390 and 0.0 <= out["distance_le_2"] <= 1.0
    assert out["by_walk"]["mean_entries_walked"] <= out["by_list"]["mean_entries_walked"] * 1.02
    assert out["by_walk"]["cut_at_cap"][1024] <= out["by_list"]["cut_at_cap"][1024]
    st = out["stage_a"]["512x512"]
    assert 0 <= st["settled"] <= st["rows"]


# ---- round 5: the sampled decision's estimate of the distinct sources (pgq_search.h: sample_distinct_sources) -------------

def _distinct_sources_model(src):
    """CPU model of the device estimator: 32 runs of 64 consecutive rows at evenly spaced offsets; the hash-set estimate
    (inverting E[distinct] = U (1 - (1 - 1/U)^s) on a geometric grid) for inputs in random order, the density of source
    CHANGES between adjacent rows for grouped inputs (a join's output), and for groups too long to count that way the
    measured stretch around the first row of every run (n x mean(1 / g))."""
    import math
    n = len(src)
    sample = min(n, 2048)
    runs = (sample + 63) // 64
    stride = n / runs

    def run_start(r):  # evenly spaced, shifted by a pseudo-random part of the stride (no aliasing with regular groups)
        room = int(stride) - 64
        jitter = ((((r + 1) * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF) >> 33) % room if room > 0 else 0
        return int(r * stride) + jitter

    rows, seen, changes, pairs = 0, set(), 0, 0
    for k in range(sample):
        p = min(n - 1, run_start(k >> 6) + (k & 63))
        v = src[p]
        pv = src[p - 1] if p > 0 else -1  # (inside a run: the row before; a run's first row: the row in front of the run)
        if v >= 0 and pv >= 0:
            pairs += 1
            changes += v != pv
        if v < 0:
            continue
        rows += 1
        seen.add(v)
    d, sr = len(seen), rows
    if sr < 1 or d < 1:
        est = 1.0
    elif d >= sr - 0.5:
        est = float(n)
    else:
        est = float(n)
        for t in range(1024):
            u = d * math.exp(math.log(n / d) * t / 1023.0)
            if u * (1.0 - math.exp(sr * math.log1p(-1.0 / u))) >= d:
                est = min(float(n), math.ceil(u))
                break
    if pairs >= 32 and changes * 2 < pairs:
        groups = n * max(changes, 0.5) / pairs
        if changes < 16:
            inv = []
            for r in range(min(runs, 32)):
                p = min(n - 1, run_start(r))
                lo = hi = p
                while hi + 1 < n and src[hi + 1] == src[p]:
                    hi += 1
                while lo > 0 and src[lo - 1] == src[p]:
                    lo -= 1
                inv.append(1.0 / (hi - lo + 1))
            groups = n * sum(inv) / len(inv)
        est = min(float(n), max(float(d), groups))
    return est


def test_distinct_source_estimate_on_grouped_and_shuffled_inputs():
    rng = np.random.default_rng(12)
    V = 448626
    for sources, per in ((2048, 32), (2048, 128), (2048, 1024), (32, 65536), (40, 1000), (10000, 7)):
        s = np.repeat(rng.choice(V, sources, replace=False), per)  # grouped by source, like a nested-loop join emits it
        est = _distinct_sources_model(s)
        assert sources / 1.6 <= est <= sources * 1.6, (sources, per, est)
    for sources, n in ((2048, 65536), (300, 100000), (50000, 65536)):  # the same sources in random order
        s = rng.choice(rng.choice(V, sources, replace=False), n)
        est = _distinct_sources_model(s)
        true = len(np.unique(s))
        assert true / 2.0 <= est <= true * 2.0, (sources, n, est, true)
    s = rng.permutation(V)[:65536]  # every row its own source
    assert _distinct_sources_model(s) == 65536.0
    # ragged groups (a join with a filter): lengths between 1 and 400
    lens = rng.integers(1, 400, 3000)
    s = np.repeat(rng.choice(V, 3000, replace=False), lens)
    est = _distinct_sources_model(s)
    assert 3000 / 1.7 <= est <= 3000 * 1.7, est


# ---- round 5: levels enqueued ahead of the host (pgq_msbfs.hip: run_batches, k_level_reset) ------------------------------

def _decide_level(rule, fe, fw, fv, unresolved, nzw):
    """decide_level of pgq_search.h: bit 0 top-down, bit 1 sparse bottom-up, bit 2 probe before the level."""
    E, V, wd = rule["E"], rule["V"], rule["wd"]
    push = fe * rule["push_div"] < E
    wpn = fe / max(E, 1.0) * (fw / max(fv, 1)) * (nzw / wd)
    sparse = (not push) and wpn < rule["sparse_below"]
    probe = False
    if rule["use_probe"]:
        probe_bytes = unresolved * (E / max(V, 1.0)) * 256.0
        level_bytes = fe * 20.0 if push else (E * (8.0 + 6.0 * wd) if not sparse else E * 4.0 + fe * 16.0 + V * (4.0 + 24.0 * wd))
        probe = probe_bytes <= level_bytes
    return (1 if push else 0) | (2 if sparse else 0) | (4 if probe else 0)


class _LaneBatchModel:
    """One lane batch as the level kernels see it: per-lane frontiers over a small graph, rows (lane, dst), the counter
    block, and `done`.  A level = [probe: answers rows at distance t from frontier t-1] expansion [detection unless probed];
    with a probe the expansion is skipped when at most `stop` rows are left open (they are deferred)."""

    def __init__(self, off, adj, srcs, rows, stop):
        self.off, self.adj, self.V = off, adj, len(off) - 1
        self.L = len(srcs)
        self.front = [{s} for s in srcs]
        self.seen = [{s} for s in srcs]
        self.rows, self.res = rows, [-1] * len(rows)
        self.stop, self.done, self.t_ran = stop, 0, 0
        self.cnt = self._counters(len(rows))

    def _counters(self, unresolved):
        verts = set().union(*self.front) if self.front else set()
        fe = sum(self.off[v + 1] - self.off[v] for f in self.front for v in f)  # per (vertex, lane): the model's own measure
        act = {l for (l, d), r in zip(self.rows, self.res) if r == -1}
        return {"fe": fe, "fv": len(verts), "fw": sum(len(f) for f in self.front), "unresolved": unresolved,
                "nzw": max(1, len(act))}

    def level(self, t, bits):
        """The kernels of level t (they return at once when `done`)."""
        if self.done:
            return
        probe = bool(bits & 4)
        if probe:  # rows at distance t: an in-neighbour of dst is in frontier t-1  <=>  dst in N_out(frontier)
            for i, (l, d) in enumerate(self.rows):
                if self.res[i] == -1 and any(d in self.adj[self.off[v]:self.off[v + 1]] for v in self.front[l]):
                    self.res[i] = t
            self.cnt["unresolved"] = sum(r == -1 for r in self.res)
            if self.cnt["unresolved"] <= self.stop:
                self.t_ran = t
                return  # the expansion returns at once; the host (or the next k_level_reset) ends the batch
        nxt = []
        for l in range(self.L):
            reach = {int(n) for v in self.front[l] for n in self.adj[self.off[v]:self.off[v + 1]]} - self.seen[l]
            self.seen[l] |= reach
            nxt.append(reach)
        self.front = nxt
        if not probe:
            for i, (l, d) in enumerate(self.rows):
                if self.res[i] == -1 and d in self.front[l]:
                    self.res[i] = t
        self.cnt = self._counters(sum(r == -1 for r in self.res))
        self.t_ran = t


def _round_trip(m, rule):
    ran, t = [], 1
    while m.cnt["unresolved"] > 0 and m.cnt["fe"] > 0:
        bits = _decide_level(rule, m.cnt["fe"], m.cnt["fw"], m.cnt["fv"], m.cnt["unresolved"], rule["wd"] if t == 1 else m.cnt["nzw"])
        m.level(t, bits)
        ran.append(bits)
        if (bits & 4) and m.cnt["unresolved"] <= m.stop:
            break
        t += 1
    return ran


def _enqueued_ahead(m, rule, plan):
    """The chain: k_level_reset(t) in front of every planned level + one behind the last; then the host's replay and, when
    the device called the levels off, the round-trip loop from there.  Returns the levels that really ran."""
    log, status, prev_stop = {}, None, -1
    for k, bits in enumerate(list(plan) + [0x80]):
        t = k + 1
        if not m.done:  # k_level_reset(t)
            log[t - 1] = dict(m.cnt)
            c = m.cnt
            code = 0
            if c["unresolved"] == 0 or c["fe"] == 0:
                code = 1
            elif prev_stop >= 0 and c["unresolved"] <= prev_stop:
                code = 1
            elif bits == 0x80:
                code = 3
            elif _decide_level(rule, c["fe"], c["fw"], c["fv"], c["unresolved"], rule["wd"] if t == 1 else c["nzw"]) != bits:
                code = 2
            if code:
                m.done, status = code, (code, t)
        if bits != 0x80:
            m.level(t, bits)
            prev_stop = m.stop if (bits & 4) else -1
    code, t_stop = status
    ran, over = [], False
    for k in range(1, t_stop):  # the host's replay of its bookkeeping from the log
        ran.append(plan[k - 1])
        if (plan[k - 1] & 4) and log[k]["unresolved"] <= m.stop:
            over = True
            break
    if not over and code != 1:  # the host takes over at level t_stop with the logged counters (= the model's own)
        assert log[t_stop - 1] == m.cnt
        m.done = 0
        t = t_stop
        while m.cnt["unresolved"] > 0 and m.cnt["fe"] > 0:
            bits = _decide_level(rule, m.cnt["fe"], m.cnt["fw"], m.cnt["fv"], m.cnt["unresolved"], rule["wd"] if t == 1 else m.cnt["nzw"])
            m.level(t, bits)
            ran.append(bits)
            if (bits & 4) and m.cnt["unresolved"] <= m.stop:
                break
            t += 1
    return ran, code


def test_levels_enqueued_ahead_protocol_matches_the_round_trip_loop():
    """Whatever plan a batch is enqueued under — the right one, a truncated one, one with wrong levels, one that is too
    long — the levels that really run and every row's answer are those of the round-trip loop."""
    rng = np.random.default_rng(31)
    codes = set()
    for trial in range(60):
        V = int(rng.integers(20, 120))
        E = int(rng.integers(V, 6 * V))
        s, d = rng.integers(0, V, E), rng.integers(0, V, E)
        order = np.argsort(s, kind="stable")
        off = np.zeros(V + 1, dtype=np.int64)
        np.cumsum(np.bincount(s, minlength=V), out=off[1:])
        adj = d[order]
        srcs = [int(x) for x in rng.choice(V, int(rng.integers(1, 12)), replace=False)]
        rows = [(int(rng.integers(0, len(srcs))), int(rng.integers(0, V))) for _ in range(int(rng.integers(1, 80)))]
        rows = [(l, dd) for l, dd in rows if dd != srcs[l]]
        if not rows:
            continue
        stop = int(rng.integers(0, 4))
        rule = {"E": float(E), "V": float(V), "wd": 1, "push_div": float(rng.choice([2.0, 12.0, 24.0])), "sparse_below": 1.5,
                "use_probe": int(rng.integers(0, 2))}
        ref = _LaneBatchModel(off, adj, srcs, rows, stop)
        want_ran = _round_trip(ref, rule)
        for variant in range(4):
            plan = list(want_ran)
            if variant == 1 and plan:
                plan = plan[:int(rng.integers(0, len(plan)))]                    # runs out
            elif variant == 2 and plan:
                plan[int(rng.integers(0, len(plan)))] ^= int(rng.choice([1, 2, 4]))  # a level the counters do not call for
            elif variant == 3:
                plan = plan + [int(rng.integers(0, 8)) for _ in range(3)]          # longer than the batch
            if not plan:
                continue
            m = _LaneBatchModel(off, adj, srcs, rows, stop)
            ran, code = _enqueued_ahead(m, rule, plan)
            codes.add(code)
            assert ran == want_ran, (trial, variant, plan, ran, want_ran)
            assert m.res == ref.res
    assert codes >= {1, 2, 3}


# ---- round 6: the source-centric kernel (pgq_ball.h) ---------------------------------------------------------------------

def ball_segments_model(src):
    """k_ball_segments: a segment starts where the source changes or at a multiple of 1024 rows."""
    n = len(src)
    starts = [i for i in range(n) if i % 1024 == 0 or src[i] != src[i - 1]]
    return [(a, (starts[k + 1] if k + 1 < len(starts) else n)) for k, a in enumerate(starts)]


def ball_model(V, off, adj, roff, radj, s, dsts, ball_cap=None, test_cap=None, rng=None):
    """k_src_ball for one segment (source s, destinations dsts): the answers it writes, None where it leaves the row open.
    ball_cap: the two-hop walk is cut after that many entries (S2 incomplete); test_cap: the backward walk of a distance-4
    candidate is cut.  The head layout (entries 0..30 / 31..62 / the rest of the list) only orders the tests: a row is
    answered 3 iff ANY in-neighbour is in S2, which the three passes together cover."""
    out = []
    if s < 0:
        return [-1] * len(dsts)
    nb1 = adj[off[s]:off[s + 1]]
    S1 = set(nb1.tolist())
    S2 = set(S1)
    walked, cut = 0, False
    order = list(nb1.tolist())
    if rng is not None:
        rng.shuffle(order)  # 16 wavefronts share the rounds: no particular order
    for v in order:
        lst = adj[off[v]:off[v + 1]].tolist()
        if ball_cap is not None and walked + len(lst) > ball_cap:
            S2.update(lst[:max(0, ball_cap - walked)])
            cut = True
            break
        walked += len(lst)
        S2.update(lst)
    for d in dsts:
        if d == s:
            out.append(0)
            continue
        if len(nb1) == 0 or roff[d + 1] == roff[d]:
            out.append(-1)
            continue
        if d in S1:
            out.append(1)
            continue
        if d in S2:  # holds for an incomplete S2 as well: S1 is complete, so d is not at distance 1
            out.append(2)
            continue
        if cut:
            out.append(None)
            continue
        ins = radj[roff[d]:roff[d + 1]].tolist()
        head1, head2, rest = ins[:31], ins[31:62], ins[60:]  # first line, second line, the list from group 15 on
        if any(u in S2 for u in head1) or any(u in S2 for u in head2) or any(u in S2 for u in rest):
            out.append(3)
            continue
        found, scanned, capped = False, 0, False
        for u in ins:
            lst = radj[roff[u]:roff[u + 1]].tolist()
            if test_cap is not None and scanned + len(lst) > test_cap:
                capped = True
                lst = lst[:max(0, test_cap - scanned)]
            scanned += len(lst)
            if any(x in S2 for x in lst):
                found = True
                break
            if capped:
                break
        out.append(4 if found else None)
    return out


def test_source_centric_ball_rule_matches_bfs():
    """The distances k_src_ball writes are BFS distances (iterativelength.cpp:34-143), with complete and with cut balls and
    walks; what it leaves open is at distance >= 5, unreachable, or behind a cap; the segments cover every row once."""
    rng = np.random.default_rng(66)
    for trial in range(12):
        V = int(rng.integers(30, 400))
        E = int(V * rng.uniform(1.0, 6.0))
        s = (rng.random(E) ** (1 + trial % 3) * V).astype(np.int64)
        d = rng.integers(0, V, E)
        off, adj, roff, radj, _, _, _ = _csr(V, s, d)
        ora = OracleCSR.from_edges(V, s, d, np.arange(E, dtype=np.int64))
        runs = rng.integers(1, 40, 12)
        srcs = rng.integers(0, V, len(runs))
        ps = np.concatenate([np.full(r, x, dtype=np.int64) for x, r in zip(srcs, runs)])
        pd = rng.integers(0, V, len(ps))
        segs = ball_segments_model(ps)
        assert sorted(i for a, b in segs for i in range(a, b)) == list(range(len(ps)))
        assert all(len(set(ps[a:b].tolist())) == 1 and b - a <= 1024 for a, b in segs)
        oln, ook = ora.lean_iterativelength(V, ps, pd)
        for caps in ((None, None), (int(rng.integers(1, 50)), None), (None, int(rng.integers(1, 30)))):
            for a, b in segs:
                got = ball_model(V, off, adj, roff, radj, int(ps[a]), pd[a:b].tolist(), caps[0], caps[1], rng)
                for k, g in enumerate(got):
                    want = int(oln[a + k]) if ook[a + k] else -1
                    if g is None:
                        assert want == -1 or want >= 3 or caps != (None, None)
                        if caps == (None, None):
                            assert want == -1 or want >= 5
                    else:
                        assert g == want, (trial, caps, int(ps[a]), int(pd[a + k]), g, want)


# ---- the two-ended batched relaxation (relax_batches_bidir, pgq_cheapest.hip), one lane ---------------------------------

def two_ended_model(V, off, adj, w, roff, radj, rw, s, t, cap0, step, rng):
    """One (src, dst) pair under relax_batches_bidir's schedule: a round = one launch forward, then one backward; a launch
    reads the pair's bound mu ONCE (its start), expands what is dirty with a label below min(cap, mu), walks the expanded
    vertex's weight-sorted list while the candidate stays below min(mu, 2 cap) (a candidate the cap cut, not mu, keeps the
    side alive), offers label + other side's label for every vertex it expands; the offers land at the launch's end.  A phase
    ends with a round in which neither side expanded: finished when mu < 2 cap or a side is not alive; else the cap rises
    and every labelled vertex is dirty again.  The order inside a launch is arbitrary (shuffled here)."""
    if s == t:
        return 0
    lists = []
    for o, a, ww in ((off, adj, w), (roff, radj, rw)):
        per = []
        for v in range(V):
            es = sorted(zip(ww[o[v]:o[v + 1]].tolist(), a[o[v]:o[v + 1]].tolist()))
            per.append(es)
        lists.append(per)
    lab = [{s: 0}, {t: 0}]
    dirty = [{s}, {t}]
    mu, cap = INF, cap0
    for _phase in range(10000):
        alive = [False, False]
        min_def = INF
        while True:
            expanded = 0
            for side in (0, 1):
                mu0 = mu  # the launch's bound
                bound = min(mu0, 2 * cap)
                cur = list(dirty[side])
                rng.shuffle(cur)
                nxt = set()
                offers = INF
                for v in cur:
                    dv = lab[side][v]
                    if dv >= mu0:
                        continue  # dead: nothing beyond the pair's bound matters
                    if dv >= cap:
                        nxt.add(v)  # deferred: over the cap
                        alive[side] = True
                        min_def = min(min_def, dv)
                        continue
                    expanded += 1
                    if v in lab[side ^ 1]:
                        offers = min(offers, dv + lab[side ^ 1][v])
                    for ww_, u in lists[side][v]:
                        cand = dv + ww_
                        if cand >= bound:
                            if cand < mu0:  # the cap cut it, not the pair's bound
                                alive[side] = True
                                min_def = min(min_def, cand >> 1)
                            break
                        if cand < lab[side].get(u, INF):
                            lab[side][u] = cand
                            nxt.add(u)
                dirty[side] = nxt
                mu = min(mu, offers)
            if expanded == 0:
                break
        if mu < 2 * cap or not alive[0] or not alive[1]:
            return None if mu == INF else mu
        nc = max(cap + step, cap + cap // 4)
        if min_def != INF and min_def + 1 > nc:
            nc = min_def + 1
        cap = nc
        dirty = [set(lab[0]), set(lab[1])]
    raise AssertionError("the schedule does not terminate")


def test_two_ended_relaxation_schedule_matches_dijkstra():
    rng = np.random.default_rng(23)
    for trial in range(60):
        V = int(rng.integers(5, 60))
        E = int(rng.integers(V, V * 6))
        s, d = rng.integers(0, V, E), rng.integers(0, V, E)
        if trial % 4 == 3:
            w = np.full(E, 7)  # one weight: nearly every band is empty
        else:
            w = rng.integers(0 if trial % 3 == 0 else 1, 40, E)  # every third graph has zero-weight edges
        off, adj, roff, radj, ww, rw, order = _csr(V, s, d, w)
        ora = OracleCSR.adopt(V, off, adj, np.arange(E, dtype=np.int64), ww.astype(np.int64))
        ps, pd = rng.integers(0, V, 30), rng.integers(0, V, 30)
        out, ok = ora.lean_cheapest_path_length(V, ps, pd)
        for a, b, want, k in zip(ps.tolist(), pd.tolist(), out.tolist(), ok.tolist()):
            cap0, step = int(rng.integers(1, 30)), int(rng.integers(1, 10))
            assert two_ended_model(V, off, adj, ww, roff, radj, rw, a, b, cap0, step, rng) == (want if k else None), (trial, a, b, cap0, step)
