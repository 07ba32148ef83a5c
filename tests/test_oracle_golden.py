"""Pins the CPU oracle against every golden vector the reference's own tests hold for the hot path
(SURVEY.md §8c).  CPU only."""
import numpy as np
import pytest

from helpers import all_pairs, directed_rows, load_golden, undirected_rows
from oracle.pgq_oracle import OracleCSR, OracleError


def build(V, rows, w=None):
    s, d, e = rows
    return OracleCSR.from_edges(V, s, d, e, w)


def test_csr_layout_getpgschema():
    g = load_golden("student_csr_layout.json")  # getpgschema.test:85-107
    c = build(g["V"], directed_rows(g["edges"]))
    assert c.v.tolist() == g["csr_v"] + [g["csr_v"][-1]] * (g["V"] + 2 - len(g["csr_v"]))
    assert c.e.tolist() == g["csr_e"]
    src = np.asarray(g["edges"])[:, 0]
    assert c.edge_ids.tolist() == np.argsort(src, kind="stable").tolist()  # slot order = arrival order per vertex


def test_csr_segfault_counts():
    g = load_golden("csr_segfault.json")  # csr_segfault.test:48-56
    V = g["V"]
    ids = np.arange(V, dtype=np.int64)
    c = build(V, (ids, ids.copy(), ids.copy()))
    assert len(c.v) == g["count_v"] and len(c.e) == g["count_e"]


def test_non_unique_vertices_error():
    # non-unique-vertices.test:40-46: edge_size != edge_size_count -> Constraint Error
    c = OracleCSR()
    c.create_csr_vertex(2, [0, 1], [1, 0])
    with pytest.raises(OracleError, match="Non-existent/non-unique vertices"):
        c.create_csr_edge(2, 1, 2, [0], [1], [0])


def test_w_type():
    g = load_golden("csr_w_type.json")  # get_csr_w_type.test:85-150
    lay = load_golden("student_csr_layout.json")
    s, d, e = directed_rows(lay["edges"])
    want = {c_["query"]: c_["value"] for c_ in g["cases"]}
    assert build(5, (s, d, e)).w_type == want["SELECT csr_get_w_type(0);"]
    assert build(5, (s, d, e), w=np.full(len(s), 12, dtype=np.int64)).w_type == want["SELECT csr_get_w_type(1);"]
    assert build(5, (s, d, e), w=np.full(len(s), 1.2)).w_type == want["SELECT csr_get_w_type(2);"]


def test_student_directed_paths():
    g = load_golden("student_directed.json")  # shortest_path.test:59-82
    V = g["V"]
    c = build(V, directed_rows(g["edges"]))
    s, d = all_pairs(V)
    ln, ok = c.iterativelength(V, s, d)
    paths = c.shortestpath(V, s, d)
    got = {}
    for i in range(len(s)):
        if ok[i] and g["lower"] <= ln[i] <= g["upper"]:
            got[(int(s[i]), int(d[i]))] = (int(ln[i]), paths[i])
    want = {(p["src"], p["dst"]): (p["length"], p["path"]) for p in g["paths"]}
    assert got == want
    for i in range(len(s)):  # path_length(p) = len(path)//2 (match.cpp:745-757)
        if paths[i] is not None:
            assert len(paths[i]) // 2 == (ln[i] if ok[i] else -1) or s[i] == d[i]
    u = g["udf_sql"]  # shortest_path.test:96-128
    got_u = sorted((k[1], v[1]) for k, v in got.items() if k[0] == u["src_filter"])
    assert got_u == sorted((p["dst"], p["path"]) for p in u["paths"])


def test_student_undirected_lengths():
    g = load_golden("student_undirected.json")  # undirected_paths.test:91-123
    V = g["V"]
    c = build(V, undirected_rows(g["edges"]))
    s, d = all_pairs(V)
    for variant in (1, 2):
        ln, ok = c.iterativelength(V, s, d, variant=variant)
        got = [[int(a), int(b), int(l)] for a, b, l, k in zip(s, d, ln, ok) if k]
        assert got == g["all_pairs"]
    ln, ok = c.iterativelength(V, s, d)
    got = {(int(a), int(b)): int(l) for a, b, l, k in zip(s, d, ln, ok) if k}
    for key in ("from0", "from4"):
        for a, b, l in g[key]["rows"]:
            assert got[(a, b)] == l
    assert sorted([a, b, l] for (a, b), l in got.items() if a == 0 and 0 <= l <= 2) == g["bounded_0_2_from0"]["rows"]
    # shortestpath lengths agree (edge ids of undirected CSRs are unspecified in the reference)
    paths = c.shortestpath(V, s, d)
    for i in range(len(s)):
        assert len(paths[i]) // 2 == got[(int(s[i]), int(d[i]))]


def test_edgeless_graph():
    g = load_golden("edgeless.json")  # edgeless_graph.test:26-34
    V = g["V"]
    c = build(V, directed_rows(np.zeros((0, 2), dtype=np.int64)))
    s, d = all_pairs(V)
    ln, ok = c.iterativelength(V, s, d)
    paths = c.shortestpath(V, s, d)
    rows = [{"src": int(a), "dst": int(b), "path": p, "length": int(l)} for a, b, l, k, p in zip(s, d, ln, ok, paths)
            if k]
    assert [(r["src"], r["dst"], r["path"], r["length"]) for r in rows] == \
        [(r["src_id"] - 1, r["dst_id"] - 1, r["path"], r["length"]) for r in g["rows"]]
    assert all(p is None for a, b, p in zip(s, d, paths) if a != b)


def test_all_properties_vertices():
    g = load_golden("all_properties_vertices.json")  # all_properties.test:69-80
    V = len(g["student_ids"])
    c = build(V, directed_rows(g["edges"]))
    s, d = all_pairs(V)
    paths = c.shortestpath(V, s, d)
    got = sorted((int(a), int(b), p[0::2]) for a, b, p in zip(s, d, paths) if p is not None)
    assert got == sorted((r["src_id"], r["dst_id"], r["vertices"]) for r in g["rows"])


def test_snb003_paths():
    g = load_golden("snb003_knows.json")  # complex_matching.test:329-360, :114-200; snb.test:108-114
    V = g["V"]
    c = build(V, directed_rows(g["edges"]))
    for key, src in (("from16_1_3", 16), ("from4_1_3", 4)):
        exp = g[key] if isinstance(g[key], list) else g[key]["paths"]
        d = np.arange(V, dtype=np.int64)
        s = np.full(V, src, dtype=np.int64)
        ln, ok = c.iterativelength(V, s, d)
        paths = c.shortestpath(V, s, d)
        got = sorted((int(b), p) for b, l, k, p in zip(d, ln, ok, paths) if k and 1 <= l <= 3)
        want = sorted((p["dst"], p["path"]) for p in exp)
        if key == "from4_1_3":  # that query joins hasInterest: only destinations with interests appear
            got = [x for x in got if x[0] in {w[0] for w in want}]
        assert got == want
        lean = c.lean_shortestpath(V, s, d)
        assert lean == paths
    ic = g["ic13_directed"]
    ln, ok = c.iterativelength(V, [ic["src"]], [ic["dst"]])
    assert ok[0] and ln[0] == ic["length"]
    cu = build(V, undirected_rows(g["edges"]))
    ic = g["ic13_undirected"]  # snb_inheritance.test:88-93
    ln, ok = cu.iterativelength(V, [ic["src"]], [ic["dst"]])
    assert ok[0] and ln[0] == ic["length"]


def test_null_and_selection_semantics():
    # iterativelength.cpp:99-103: NULL src -> NULL(-1); src==dst -> 0 without a lane; dst validity ignored.
    g = load_golden("student_directed.json")
    V = g["V"]
    c = build(V, directed_rows(g["edges"]))
    src = np.array([0, 4, 2, 2, 1], dtype=np.int64)
    dst = np.array([3, 0, 2, 0, 4], dtype=np.int64)
    valid = np.array([True, False, True, True, True])
    ln, ok = c.iterativelength(V, src, dst, src_valid=valid)
    assert ln.tolist() == [1, -1, 0, 2, -1] and ok.tolist() == [True, False, True, True, False]
    paths = c.shortestpath(V, src, dst, src_valid=valid)
    assert paths == [[0, 2, 3], None, [2], [2, 6, 3, 3, 0], None]
    # dictionary/constant vectors arrive as (data, sel): constant src = sel of zeros
    sel = np.zeros(5, dtype=np.uint32)
    ln, ok = c.iterativelength(V, np.array([4], dtype=np.int64), np.arange(5, dtype=np.int64), src_sel=sel)
    assert [int(l) if k else None for l, k in zip(ln, ok)] == [2, 3, 3, 1, 0]


def test_more_than_512_pairs_batches():
    # > LANE_LIMIT rows exercise the batch loop (iterativelength.cpp:84); literal == lean on a random graph
    rng = np.random.default_rng(7)
    V, E = 300, 1500
    s, d = rng.integers(0, V, E), rng.integers(0, V, E)
    c = build(V, (s, d, np.arange(E, dtype=np.int64)))
    ps, pd = rng.integers(0, V, 1300), rng.integers(0, V, 1300)
    ln, ok = c.iterativelength(V, ps, pd)
    ln2, ok2 = c.iterativelength(V, ps, pd, variant=2)
    lln, lok = c.lean_iterativelength(V, ps, pd, nthreads=2)
    assert (ok == lok).all() and (ln[ok] == lln[ok]).all()
    assert (ok2 == ok).all() and (ln2 == ln).all()
    assert c.shortestpath(V, ps, pd) == c.lean_shortestpath(V, ps, pd)
    bl, bok = c.baseline_run("iterativelength", V, ps, pd, nthreads=3)
    assert (bok == ok).all() and (bl == ln).all()
    bl, bok = c.baseline_run("shortestpath", V, ps[:600], pd[:600], nthreads=2)
    assert (bok == ok[:600]).all() and (bl[bok] == ln[:600][bok]).all()


def test_cheapest_path_literal_vs_lean():
    # No reference test covers cheapest_path_length (parity unpinned): literal Bellman-Ford restatement
    # (cheapest_path_length.cpp:52-136) must equal an independent Dijkstra, int64 and double weights.
    rng = np.random.default_rng(11)
    V, E = 200, 900
    s, d = rng.integers(0, V, E), rng.integers(0, V, E)
    for w in (rng.integers(1, 1000, E), rng.random(E) + 0.01):
        c = build(V, (s, d, np.arange(E, dtype=np.int64)), w=w)
        for n in (1, 3, 70, 300):  # exercises the 256/128/64/16/8/4/2/1 ladder (:101-135)
            ps, pd = rng.integers(0, V, n), rng.integers(0, V, n)
            out, ok = c.cheapest_path_length(V, ps, pd)
            lout, lok = c.lean_cheapest_path_length(V, ps, pd)
            assert (ok == lok).all()
            assert (out[ok] == lout[ok]).all()  # bit-exact, also for doubles
            assert (out[ps == pd] == 0).all()
    rep = load_golden("snb003_replyof.json")
    e = np.asarray(rep["edges"], dtype=np.int64)
    w = rng.integers(1, 50, len(e))
    c = build(rep["V"], (e[:, 0], e[:, 1], np.arange(len(e), dtype=np.int64)), w=w)
    ps, pd = e[:64, 0], e[:64, 1]
    out, ok = c.cheapest_path_length(rep["V"], ps, pd)
    assert ok.all() and (out == w[:64]).all()  # forest: the only path child->parent is the edge itself


@pytest.mark.parametrize("seed", range(6))
def test_fuzz_literal_vs_lean_multigraphs(seed):
    """The GPU parity tests at scale lean on the *lean* oracle; this pins it against the *literal* restatement
    (the one checked against the reference's golden vectors) on random multigraphs with parallel edges, self loops,
    isolated vertices, a heavy source, NULL rows and zero weights — lengths, full paths, cheapest distances."""
    rng = np.random.default_rng(900 + seed)
    V = int(rng.integers(2, 260))
    E = int(rng.integers(1, 5 * V))
    s, d = rng.integers(0, V, E), rng.integers(0, V, E)
    s[: E // 8] = s[0]
    d[E // 2: E // 2 + E // 10] = s[E // 2: E // 2 + E // 10]  # self loops
    eid = rng.permutation(E).astype(np.int64)
    n = int(rng.integers(1, 1400))
    ps, pd = rng.integers(0, V, n), rng.integers(0, V, n)
    valid = rng.random(n) > 0.1
    c = build(V, (s, d, eid))
    ln, ok = c.iterativelength(V, ps, pd, src_valid=valid)
    lln, lok = c.lean_iterativelength(V, ps, pd)
    assert (ok == (lok & valid)).all() and (ln[ok] == lln[ok]).all()
    assert c.shortestpath(V, ps[:520], pd[:520]) == c.lean_shortestpath(V, ps[:520], pd[:520])
    w = rng.integers(0, 20, E) if seed % 2 == 0 else np.round(rng.random(E) * 4, 2)
    cw = build(V, (s, d, eid), w=w)
    out, cok = cw.cheapest_path_length(V, ps[:300], pd[:300])
    lout, lcok = cw.lean_cheapest_path_length(V, ps[:300], pd[:300])
    assert (cok == lcok).all() and (out[cok] == lout[cok]).all()


def test_analytics_goldens_lcc_pagerank_wcc():
    """The other CSR consumers: literal restatements against the reference's goldens."""
    lcc = load_golden("lcc.json")
    g = lcc["student"]
    ora = OracleCSR.from_edges(g["V"], *undirected_rows(g["edges"]))  # CreateUndirectedCSRCTE feed
    got = ora.local_clustering_coefficient(np.arange(g["V"]))
    assert [repr_float32(x) for x in got] == [r[1] for r in g["rows"]]
    snb = load_golden("snb003_knows.json")
    ora = OracleCSR.from_edges(snb["V"], *undirected_rows(snb["edges"]))
    ids = [r[0] for r in lcc["snb003"]["rows"]]
    got = ora.local_clustering_coefficient(np.array(ids))
    assert [repr_float32(x) for x in got] == [r[1] for r in lcc["snb003"]["rows"]]
    pr = load_golden("pagerank.json")
    for key in ("g1", "g2"):
        g = pr[key]
        ora = OracleCSR.from_edges(g["V"], *directed_rows(g["edges"]))
        rank, it = ora.pagerank()
        assert len(rank) == g["V"] + 2 and it > 1
        for vid, txt in g["rows"]:
            assert abs(rank[vid] - float(txt)) <= 1e-15 * max(1.0, abs(float(txt))), (key, vid, rank[vid], txt)
    for case in load_golden("wcc.json")["cases"]:
        ora = OracleCSR.from_edges(case["V"], *undirected_rows(case["edges"]))
        out, ok = ora.weakly_connected_component(np.arange(case["V"]))
        assert ok.all() and [[i, int(c)] for i, c in enumerate(out)] == case["rows"], case["source"]


def repr_float32(x):
    """DuckDB's FLOAT rendering: the shortest decimal that round-trips the float32."""
    s = np.format_float_positional(np.float32(x), unique=True, trim="0")
    return s if "." in s else s + ".0"
