"""GPU parity: the HIP path (through the C ABI) against the CPU oracle and the reference's golden vectors.
Bit-exact: hop counts, NULL masks, full [v,e,v,...] paths, int64 and double cheapest-path distances."""
import numpy as np
import pytest

import duckpgq_extension_amd as pgq
from duckpgq_extension_amd import graphgen
from helpers import all_pairs, directed_rows, load_golden, undirected_rows
from oracle.pgq_oracle import OracleCSR

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _defaults():
    for k, v in (("words", 0), ("force_mode", 0), ("push_div", 12), ("hub_chunk", 4096), ("push_chunk", 256), ("probe", 1), ("defer", 8), ("force_pull", 0), ("sparse_lds", 1), ("streams", 2), ("sparse_pw", 1), ("sparse_unroll", 2), ("sparse_spill", 3),
                 ("relax_small_limit", 2048), ("chain", 1), ("chain_cap", 4096), ("probe2", 1), ("probe2_abs", 512), ("lanes", 1), ("lanes_unroll", 2),
                 # the pair-centric pre-pass would answer most pairs of these small graphs before the level kernels
                 # under test see them; the tests that exercise it switch it on themselves
                 ("meet", 0), ("meet_cap", 1 << 14), ("meet_cap_paths", 1 << 14), ("meet_cap_small", 1 << 14), ("meet_small_rows", 16384), ("meet_wide_rows", 2048), ("meet_wide_rows_always", 0), ("meet_spin_wait", 0), ("paths_reserve_mb", 1024), ("chunk_zero_copy", 1), ("meet_bias", 1.0), ("meet4", 1), ("meet4_cap", 1 << 20), ("meet4_lds_kb", 150),
                 ("bibfs_rows", 256), ("bibfs_cap", 8 << 20), ("bibfs_queue", 1 << 17),
                 # the per-row weighted search would answer every int64 row before the relaxation kernels under test run
                 ("wbibfs", 0), ("wbibfs_cap", 64 << 20), ("wbibfs_queue", 1 << 17), ("wbibfs_far", 1 << 21), ("wbibfs_delta_div", 64), ("wbibfs_mem_mb", 2048),
                 ("meet_layout", 1), ("meet_align", 4), ("probe_always", 0), ("meet_grid_mult", 8), ("meet4_grid_mult", 2), ("relax_delta_div", 0), ("relax_light", 2), ("relax_light_div", 4), ("relax_split", 1), ("relax_labels32", 1),
                 # the two-ended relaxation (round 6) would take every list of int64-weighted pairs before the one-sided kernels under
                 # test see them; its own test and the shipped configuration switch it on
                 ("relax_bidir", 0), ("relax_bidir_rows", 2), ("relax_bidir_c0_div", 64), ("relax_bidir_step_div", 128),
                 ("spec_levels", 1), ("sort_single_batch", 0), ("detect_unroll", 4), ("detect_grid_mult", 8), ("route_memo", 1), ("stage2_ahead", 1), ("meet_calibrate", 1),
                 # the source-centric kernel (round 6) would take every grouped input before the kernels under test see it; its own
                 # tests and the shipped configuration switch it on
                 ("ball", 0), ("ball_cap", 1 << 20), ("ball_test_cap", 1 << 15), ("ball_bias", 1.0), ("ball_sort", 1), ("route_timing", 1), ("route_timing_rows", 65536), ("route_try_factor", 4.0), ("calibration_cache", 1), ("ball_seg_kb", 512), ("ball_grid", 0), ("ball_head_mb", 512)):
        pgq.set_option(k, v)
    yield


SHIPPED_KEYS = ("push_div", "streams", "probe2_abs", "meet", "meet_align", "meet_bias", "meet_cap", "meet_cap_small", "meet_cap_paths",
                "meet_small_rows", "probe", "probe2", "defer", "lanes", "lanes_unroll", "sparse_lds", "sparse_pw", "sparse_unroll",
                "sparse_spill", "hub_chunk", "push_chunk", "spec_levels", "sort_single_batch", "detect_unroll", "route_memo", "meet4",
                "bibfs_rows", "relax_light", "relax_split", "relax_streams", "wbibfs", "meet4_grid_mult", "meet_grid_mult", "meet_calibrate", "stage2_ahead",
                "ball", "ball_cap", "ball_test_cap", "ball_sort", "relax_bidir")


@pytest.fixture(params=["fixture_values", "shipped_values"])
def base_config(request):
    """The tests that sweep the kernels' variants run twice: under this file's fixture values (pre-pass off, push_div 12,
    two streams, probe2_abs 512: chosen so that the level kernels see the rows) and under the values the library SHIPS
    with (read back from the library itself: pgq_get_default_option) — what a DuckDB process that sets nothing runs."""
    if request.param == "shipped_values":
        for k in SHIPPED_KEYS:
            pgq.set_option(k, pgq.get_default_option(k))
    return request.param


# The shipped configuration answers rows through the pair-centric pre-pass (and int64 weighted rows through the per-row
# search) before the lane-batched kernels see them; on tiny graphs its cost model would decline, so `meet_bias` forces it.
# Tests that replay the reference's golden vectors / fuzz tiny graphs run once per configuration.
PRODUCT_CONFIGS = {
    "lane_batches": {"meet": 0, "wbibfs": 0},
    "prepass": {"meet": 1, "meet_bias": 1e9, "wbibfs": 0},
    "prepass+wbibfs": {"meet": 1, "meet_bias": 1e9, "wbibfs": 1},
}


@pytest.fixture(params=sorted(PRODUCT_CONFIGS))
def product_config(request):
    for k, v in PRODUCT_CONFIGS[request.param].items():
        pgq.set_option(k, v)
    return request.param


def both(V, rows, w=None, csr_id=0):
    s, d, e = rows
    st = pgq.PgqState()
    st.build_csr(csr_id, V, s, d, e, w)
    return st, OracleCSR.from_edges(V, s, d, e, w)


def lens(out, ok):
    return [int(v) if k else None for v, k in zip(out, ok)]


# ---- golden vectors of the reference's own tests, replayed on the GPU ------------------------------------------

def test_golden_student_directed(product_config):
    g = load_golden("student_directed.json")  # shortest_path.test:59-82
    V = g["V"]
    st, _ = both(V, directed_rows(g["edges"]))
    s, d = all_pairs(V)
    ln, ok = st.iterativelength(0, V, s, d)
    paths = st.shortestpath(0, V, s, d)
    got = {(int(a), int(b)): (int(l), p) for a, b, l, k, p in zip(s, d, ln, ok, paths) if k and 1 <= l <= 3}
    assert got == {(p["src"], p["dst"]): (p["length"], p["path"]) for p in g["paths"]}


def test_golden_student_undirected_and_edgeless(product_config):
    g = load_golden("student_undirected.json")  # undirected_paths.test:91-123
    V = g["V"]
    st, _ = both(V, undirected_rows(g["edges"]))
    s, d = all_pairs(V)
    ln, ok = st.iterativelength(0, V, s, d)
    assert [[int(a), int(b), int(l)] for a, b, l, k in zip(s, d, ln, ok) if k] == g["all_pairs"]
    e = load_golden("edgeless.json")  # edgeless_graph.test:26-34
    st, _ = both(e["V"], directed_rows(np.zeros((0, 2), dtype=np.int64)))
    s, d = all_pairs(e["V"])
    ln, ok = st.iterativelength(0, e["V"], s, d)
    paths = st.shortestpath(0, e["V"], s, d)
    assert [(int(a), p, int(l)) for a, b, l, k, p in zip(s, d, ln, ok, paths) if k] == \
        [(r["src_id"] - 1, r["path"], r["length"]) for r in e["rows"]]
    assert all(p is None for a, b, p in zip(s, d, paths) if a != b)


def test_golden_snb003_paths(product_config):
    g = load_golden("snb003_knows.json")  # complex_matching.test:329-360
    V = g["V"]
    st, ora = both(V, directed_rows(g["edges"]))
    d = np.arange(V, dtype=np.int64)
    s = np.full(V, 16, dtype=np.int64)
    ln, ok = st.iterativelength(0, V, s, d)
    paths = st.shortestpath(0, V, s, d)
    got = sorted((int(b), p) for b, l, k, p in zip(d, ln, ok, paths) if k and 1 <= l <= 3)
    assert got == sorted((p["dst"], p["path"]) for p in g["from16_1_3"])
    s, d = all_pairs(V)  # all 2500 pairs against the oracle
    ln, ok = st.iterativelength(0, V, s, d)
    oln, ook = ora.iterativelength(V, s, d)
    assert lens(ln, ok) == lens(oln, ook)
    assert st.shortestpath(0, V, s, d) == ora.lean_shortestpath(V, s, d)
    ic = g["ic13_directed"]  # snb.test:108-114
    ln, ok = st.iterativelength(0, V, [ic["src"]], [ic["dst"]])
    assert ok[0] and ln[0] == ic["length"]
    st, _ = both(V, undirected_rows(g["edges"]))
    ic = g["ic13_undirected"]
    ln, ok = st.iterativelength(0, V, [ic["src"]], [ic["dst"]])
    assert ok[0] and ln[0] == ic["length"]


def test_null_selection_and_reachability(product_config):
    g = load_golden("student_directed.json")
    V = g["V"]
    st, ora = both(V, directed_rows(g["edges"]))
    src = np.array([0, 4, 2, 2, 1], dtype=np.int64)
    dst = np.array([3, 0, 2, 0, 4], dtype=np.int64)
    valid = np.array([True, False, True, True, True])
    ln, ok = st.iterativelength(0, V, src, dst, src_valid=valid)
    assert ln.tolist() == [1, -1, 0, 2, -1] and ok.tolist() == [True, False, True, True, False]
    assert st.shortestpath(0, V, src, dst, src_valid=valid) == [[0, 2, 3], None, [2], [2, 6, 3, 3, 0], None]
    sel = np.zeros(5, dtype=np.uint32)  # constant vector
    ln, ok = st.iterativelength(0, V, np.array([4], dtype=np.int64), np.arange(5, dtype=np.int64), src_sel=sel)
    assert lens(ln, ok) == [2, 3, 3, 1, 0]
    r, rok = st.reachability(0, V, src, dst, src_valid=valid)
    assert r.tolist() == [True, False, True, True, False] and rok.tolist() == [True, False, True, True, True]
    for variant in (2, 3):  # iterativelength2 / iterativelengthbidirectional surfaces: same hop counts
        ln2, ok2 = st.iterativelength(0, V, src, dst, src_valid=valid, variant=variant)
        assert (ln2 == ln_ref(ora, V, src, dst, valid)[0]).all()


def ln_ref(ora, V, src, dst, valid):
    return ora.iterativelength(V, src, dst, src_valid=valid)


# ---- random graphs: every lane-word width, both directions, hubs -------------------------------------------------

def random_graph(rng, V, E, skew=False):
    if skew:
        s = (rng.random(E) ** 3 * V).astype(np.int64)
        d = (rng.random(E) ** 2 * V).astype(np.int64)
    else:
        s, d = rng.integers(0, V, E), rng.integers(0, V, E)
    return s, d, np.arange(E, dtype=np.int64)


@pytest.mark.parametrize("words", [1, 2, 4, 8, 16, 32])
@pytest.mark.parametrize("mode", [0, 1, 2])
def test_random_graph_all_variants(words, mode, base_config):
    rng = np.random.default_rng(100 * words + mode)
    V, E = 3000, 14000
    rows = random_graph(rng, V, E, skew=True)
    st, ora = both(V, rows)
    pgq.set_option("words", words)
    pgq.set_option("force_mode", mode)
    pgq.set_option("hub_chunk", 64)  # exercise the split-vertex (hub) paths on a small graph
    pgq.set_option("push_chunk", 64)
    n = 1500
    ps, pd = rng.integers(0, V, n), rng.integers(0, V, n)
    valid = rng.random(n) > 0.05
    oln, ook = ora.lean_iterativelength(V, ps, pd)
    want = [int(v) if (k and vv) else None for v, k, vv in zip(oln, ook, valid)]
    opaths = ora.lean_shortestpath(V, ps[:700], pd[:700])
    # destination probe on / classic post-expansion detection; adaptive / sparse-only / dense-only bottom-up kernel
    # lanes: sparse levels run k_pull_lanes (lane-list records) / k_pull_sparse (packed lane-words)
    for probe, force_pull, lds, pw, unroll, spill, lanes in ((1, 0, 1, 2, 4, 2, 1), (0, 1, 0, 1, 2, 1, 0), (1, 2, 1, 2, 4, 2, 1), (0, 1, 1, 3, 4, 99, 0),
                                                            (0, 1, 1, 2, 1, 0, 0), (0, 1, 1, 1, 4, 1, 0), (0, 1, 1, 1, 4, 1, 1),
                                                            (0, 1, 0, 1, 2, 1, 1), (1, 1, 1, 1, 1, 1, 1), (1, 1, 0, 1, 4, 1, 1)):
        pgq.set_option("lanes", lanes)
        pgq.set_option("lanes_unroll", unroll)
        pgq.set_option("sparse_spill", spill)    # trips before the remaining words are spread over the wavefront
        pgq.set_option("streams", 1 + (probe + lds) % 3)  # 1..3 concurrent batch workers
        pgq.set_option("sparse_pw", pw)          # packed words per chunk per accumulate trip in k_pull_sparse
        pgq.set_option("sparse_unroll", unroll)  # 64-entry chunks in flight per wavefront
        pgq.set_option("probe", probe)
        pgq.set_option("probe_always", (lds + unroll) % 2)  # probe before every level / only where its cost model says so
        pgq.set_option("force_pull", force_pull)
        pgq.set_option("probe2", 1 - lds if probe else 1)  # two-hop destination probe on / off
        pgq.set_option("sparse_lds", lds)  # 1-bit frontier map in LDS (1024-thread groups) or in global memory
        # round 5: levels enqueued ahead of the host under the previous call's plan (the variants change the level rule
        # under it: plans that no longer fit are called off on the device) / one wait per level; rows of a one-batch call
        # left in place / sorted by lane; 1, 2 or 4 rows per thread in k_detect
        pgq.set_option("spec_levels", (lds + pw) % 2)
        pgq.set_option("sort_single_batch", (unroll >> 1) % 2)
        pgq.set_option("detect_unroll", unroll)
        pgq.set_option("stage2_ahead", (pw + probe) % 2)  # lane ids + batch start in front of the lane assignment's wait (same row count as the call before)
        ln, ok = st.iterativelength(0, V, ps, pd, src_valid=valid)
        assert lens(ln, ok) == want
        assert st.shortestpath(0, V, ps[:700], pd[:700]) == opaths


@pytest.mark.parametrize("cap,lds_kb", [(1 << 14, 150), (2000, 150), (64, 0), (1 << 14, 0)])
def test_meet_prepass_several_wavefronts_per_row(cap, lds_kb):
    # k_meet3w: chunk-sized calls (<= meet_wide_rows) on graphs beyond the Infinity Cache get 4 wavefronts per row while all rows
    # are resident at once (<= 1024), else 2; meet_wide_rows_always takes it on any graph.  Same answers as k_meet3, cut walks
    # (small caps) handed to the bit-map kernel (LDS / global maps) from their start.
    rng = np.random.default_rng(91 + cap)
    V, E = 6000, 60000
    st, ora = both(V, random_graph(rng, V, E, skew=True))
    V2 = 4000
    st2, ora2 = both(V2, random_graph(rng, V2, 5000), csr_id=1)  # sparse: dead ends, unreachable pairs, long distances
    pgq.set_option("meet", 1)
    pgq.set_option("meet_bias", 1e9)
    pgq.set_option("meet_cap_small", cap)
    pgq.set_option("meet4_lds_kb", lds_kb)
    pgq.set_option("meet_wide_rows_always", 1)
    for n in (1, 63, 900, 2000, 2048):
        ps, pd = rng.integers(0, V, n), rng.integers(0, V, n)
        ps[: n // 40] = pd[: n // 40]
        valid = rng.random(n) > 0.05
        oln, ook = ora.lean_iterativelength(V, ps, pd)
        ln, ok = st.iterativelength(0, V, ps, pd, src_valid=valid)
        assert lens(ln, ok) == [int(v) if (k and vv) else None for v, k, vv in zip(oln, ook, valid)], n
        ps, pd = rng.integers(0, V2, n), rng.integers(0, V2, n)
        oln, ook = ora2.lean_iterativelength(V2, ps, pd)
        ln, ok = st2.iterativelength(1, V2, ps, pd)
        assert lens(ln, ok) == [int(v) if k else None for v, k in zip(oln, ook)], n
    # ids outside [0, V) are refused like everywhere
    dev = st.device_csr(0)
    import torch
    t_s = torch.from_numpy(rng.integers(0, V, 500)).cuda()
    t_d = torch.from_numpy(rng.integers(0, V, 500)).cuda()
    t_o = torch.empty(500, dtype=torch.int64, device="cuda")
    dev.iterativelength_bulk_ptr(500, t_s.data_ptr(), t_d.data_ptr(), t_o.data_ptr())
    t_d[17] = V
    with pytest.raises(pgq.PgqError):
        dev.iterativelength_bulk_ptr(500, t_s.data_ptr(), t_d.data_ptr(), t_o.data_ptr())


def test_shortestpath_lists_written_again_when_the_reservation_was_short():
    # paths_reserve_mb: the pre-pass's list buffer is reserved up to a limit up front (9 elements per row fit below ~15 M rows
    # at the shipped 1 GB); lists that did not fit are written again into a buffer of the exact size.  0 MB = 4 KB forces it.
    rng = np.random.default_rng(101)
    V, E = 6000, 60000
    st, ora = both(V, random_graph(rng, V, E, skew=True))
    pgq.set_option("meet", 1)
    pgq.set_option("meet_bias", 1e9)
    ps, pd = rng.integers(0, V, 3000), rng.integers(0, V, 3000)
    want = ora.lean_shortestpath(V, ps, pd)
    for mb in (0, 1024, 0):
        pgq.set_option("paths_reserve_mb", mb)
        assert st.shortestpath(0, V, ps, pd) == want, mb
    pgq.set_option("paths_reserve_mb", 1024)


def test_meet_spin_wait_returns_complete_results():
    # meet_spin_wait = 1: the chain's wait polls the report its last workgroup writes into pinned memory; chunk calls (results in
    # the pinned staging block) and bulk calls (results in HBM, read back here through the library's stream) — 200 calls each
    import torch
    rng = np.random.default_rng(97)
    V, E = 6000, 60000
    st, ora = both(V, random_graph(rng, V, E, skew=True))
    pgq.set_option("meet", 1)
    pgq.set_option("meet_bias", 1e9)
    pgq.set_option("meet_spin_wait", 1)
    dev = st.device_csr(0)
    for n in (1, 64, 2048, 8192):
        ps, pd = rng.integers(0, V, n), rng.integers(0, V, n)
        oln, ook = ora.lean_iterativelength(V, ps, pd)
        want = [int(v) if k else None for v, k in zip(oln, ook)]
        for _ in range(50 if n <= 2048 else 10):
            ln, ok = st.iterativelength(0, V, ps, pd)
            assert lens(ln, ok) == want, n
        t_s, t_d = torch.from_numpy(ps).cuda(), torch.from_numpy(pd).cuda()
        for _ in range(50 if n <= 2048 else 10):
            t_o = torch.full((n,), -7, dtype=torch.int64, device="cuda")
            dev.iterativelength_bulk_ptr(n, t_s.data_ptr(), t_d.data_ptr(), t_o.data_ptr())
            torch.cuda.synchronize()
            assert (t_o.cpu().numpy() == np.where(ook, oln, -1)).all(), n


@pytest.mark.parametrize("cap,lds_kb,align", [(1 << 18, 150, 4), (3000, 150, 16), (1, 150, 4), (1 << 18, 0, 32), (300, 150, 4)])
def test_meet_prepass_matches_oracle(cap, lds_kb, align, base_config):
    # k_meet3 (pgq_meet.hip): distances 1..3 from two-hop scans, everything else handed to the lane-batched search.
    # cap = adjacency entries a pair's walk may scan before it is handed on: small caps leave most rows to k_meet4d / the
    # MS-BFS path (all paths mixed)
    rng = np.random.default_rng(77 + cap)
    V, E = 6000, 60000
    rows = random_graph(rng, V, E, skew=True)  # skewed: lists longer than the 512-entry hash table exist
    pgq.set_option("meet_align", align)  # read at upload: padded lists start on 16 / 64 / 128-byte boundaries
    st, ora = both(V, rows)
    pgq.set_option("meet", 1)
    pgq.set_option("meet_cap", cap)
    pgq.set_option("meet_cap_paths", cap)
    pgq.set_option("meet_bias", 1e9)  # always take the pre-pass
    pgq.set_option("meet4", 0 if cap == 1 else 1)  # k_meet4: LDS bit-map kernel for what k_meet3 leaves open
    pgq.set_option("meet4_cap", 1 << 20 if cap != 3000 else 2000)
    pgq.set_option("meet4_lds_kb", lds_kb)  # 0: the vertex bit maps of k_meet4 live in global memory (graphs over ~1.2 M vertices)
    n = 3000
    ps, pd = rng.integers(0, V, n), rng.integers(0, V, n)
    ps[:50] = pd[:50]  # src == dst rows
    valid = rng.random(n) > 0.05
    oln, ook = ora.lean_iterativelength(V, ps, pd)
    want = [int(v) if (k and vv) else None for v, k, vv in zip(oln, ook, valid)]
    pgq.reset_stats()
    ln, ok = st.iterativelength(0, V, ps, pd, src_valid=valid)
    assert lens(ln, ok) == want
    stats = pgq.get_stats()
    assert stats["meet_pairs"] > 0
    if cap == 1:
        assert stats["levels"] > 0  # the MS-BFS path answered what the pre-pass left open
    # shortestpath through the pre-pass: inner vertices by the reference's tie-break (smallest parent, first slot),
    # lists of the rows left to the lane-batched search appended behind them
    pgq.reset_stats()
    got = st.shortestpath(0, V, ps[:1500], pd[:1500], src_valid=valid[:1500])
    opaths = ora.lean_shortestpath(V, ps[:1500], pd[:1500])
    assert got == [p if vv else None for p, vv in zip(opaths, valid[:1500])]
    assert pgq.get_stats()["meet_pairs"] > 0
    # a sparse directed graph: many dead ends and unreachable pairs, long distances
    V2 = 4000
    rows2 = random_graph(rng, V2, 5000)
    st2, ora2 = both(V2, rows2, csr_id=1)
    ps, pd = rng.integers(0, V2, 2000), rng.integers(0, V2, 2000)
    oln, ook = ora2.lean_iterativelength(V2, ps, pd)
    ln, ok = st2.iterativelength(1, V2, ps, pd)
    assert lens(ln, ok) == [int(v) if k else None for v, k in zip(oln, ook)]
    assert st2.shortestpath(1, V2, ps, pd) == ora2.lean_shortestpath(V2, ps, pd)


@pytest.mark.parametrize("lds_kb", [150, 0])
def test_bibfs_few_open_rows_any_distance(lds_kb):
    # k_bibfs (pgq_meet.hip): one bidirectional search per row for the few rows the two-hop kernels leave open — long
    # distances, unreachable pairs (one side's closure is exhausted), caps that hand rows on to the lane-batched search
    rng = np.random.default_rng(4242 + lds_kb)
    pgq.set_option("meet", 1)
    pgq.set_option("meet_bias", 1e9)
    pgq.set_option("meet4_lds_kb", lds_kb)  # 0: visited maps in global memory (graphs whose maps do not fit in LDS)
    # (a) a directed ring with a few chords: distances up to the hundreds, everything reachable
    V = 3000
    ring_s = np.arange(V, dtype=np.int64)
    ring_d = (ring_s + 1) % V
    ch_s, ch_d = rng.integers(0, V, 40), rng.integers(0, V, 40)
    s, d = np.concatenate([ring_s, ch_s]), np.concatenate([ring_d, ch_d])
    st, ora = both(V, (s, d, np.arange(len(s), dtype=np.int64)))
    ps, pd = rng.integers(0, V, 150), rng.integers(0, V, 150)
    oln, ook = ora.lean_iterativelength(V, ps, pd)
    want = [int(v) if k else None for v, k in zip(oln, ook)]
    assert max(v for v in want if v is not None) > 20
    pgq.reset_stats()
    ln, ok = st.iterativelength(0, V, ps, pd)
    assert lens(ln, ok) == want
    assert pgq.get_stats()["levels"] == 0  # every row was answered before the lane-batched search
    # (b) sparse directed graph: dead ends, unreachable pairs with a small closure on one side
    V2 = 5000
    rows2 = random_graph(rng, V2, 6000)
    st2, ora2 = both(V2, rows2, csr_id=1)
    ps, pd = rng.integers(0, V2, 200), rng.integers(0, V2, 200)
    oln, ook = ora2.lean_iterativelength(V2, ps, pd)
    want2 = [int(v) if k else None for v, k in zip(oln, ook)]
    assert None in want2
    ln, ok = st2.iterativelength(1, V2, ps, pd)
    assert lens(ln, ok) == want2
    # (c) caps: expansions over `bibfs_cap` entries / frontiers over `bibfs_queue` vertices leave rows to the MS-BFS path
    for cap, queue in ((40, 1 << 17), (8 << 20, 1024)):
        pgq.set_option("bibfs_cap", cap)
        pgq.set_option("bibfs_queue", queue)
        rows3 = random_graph(rng, 4000, 40000, skew=True)
        st3, ora3 = both(4000, rows3, csr_id=2)
        pgq.set_option("meet_cap", 200)   # most rows pass the two-hop kernels unanswered ...
        pgq.set_option("meet4_cap", 200)
        ps, pd = rng.integers(0, 4000, 120), rng.integers(0, 4000, 120)
        oln, ook = ora3.lean_iterativelength(4000, ps, pd)
        ln, ok = st3.iterativelength(2, 4000, ps, pd)
        assert lens(ln, ok) == [int(v) if k else None for v, k in zip(oln, ook)]


def grouped_rows(rng, V, runs, null_run=0):
    """Rows grouped by source like a nested-loop join emits them (match.cpp:467-495): `runs` = run lengths, each with a source
    of its own (repeats allowed: a source may come back in a later run), random destinations with a few src == dst rows."""
    srcs = rng.integers(0, V, len(runs))
    ps = np.concatenate([np.full(r, s, dtype=np.int64) for s, r in zip(srcs, runs)])
    pd = rng.integers(0, V, len(ps))
    same = rng.random(len(ps)) < 0.01
    pd[same] = ps[same]
    valid = np.ones(len(ps), dtype=bool)
    if null_run:  # a stretch of NULL sources in the middle (their payload is garbage, as DuckDB leaves it)
        at = len(ps) // 2
        valid[at:at + null_run] = False
    return ps, pd, valid


@pytest.mark.parametrize("head_mb", [512, 0])
@pytest.mark.parametrize("lds_kb", [150, 0])
@pytest.mark.parametrize("ball_cap,test_cap", [(1 << 20, 1 << 15), (300, 1 << 15), (1 << 20, 40)])
def test_source_centric_ball_matches_oracle(lds_kb, ball_cap, test_cap, head_mb):
    # k_ball_segments + k_src_ball (pgq_ball.h): rows grouped by source answered from ONE two-hop ball per source run in a
    # vertex bit map (LDS, or a global slice: lds_kb = 0) — d in S1 / S2, an in-neighbour of d in S2 (3), an in-neighbour
    # of an in-neighbour (4); what it leaves open (distance >= 5, unreachable, balls / walks over their caps) goes through
    # the pre-pass and the lane batches.  Run lengths that are not multiples of 1024, runs across window boundaries, single rows.
    rng = np.random.default_rng(606 + lds_kb + ball_cap + test_cap)
    pgq.set_option("meet", 1)
    pgq.set_option("meet_bias", 1e9)
    pgq.set_option("ball", 2)  # forced: the decision has its own test
    # in-list heads at a fixed stride (pgq_csr::rhead, read at upload and at launch): a row's scan starts from its destination
    # id alone / 0: the list positions are gathered per row (graphs whose heads do not fit the budget)
    pgq.set_option("ball_head_mb", head_mb)
    pgq.set_option("ball_cap", ball_cap)
    pgq.set_option("ball_test_cap", test_cap)
    pgq.set_option("meet4_lds_kb", lds_kb)
    runs = [1, 3, 64, 700, 1024, 1500, 2500, 1, 1, 90, 2048, 5]
    # (a) skewed small-world graph: nearly everything within 4 hops, hubs whose balls run over a small cap
    V, E = 6000, 60000
    st, ora = both(V, random_graph(rng, V, E, skew=True))
    ps, pd, valid = grouped_rows(rng, V, runs, null_run=37)
    oln, ook = ora.lean_iterativelength(V, ps, pd, nthreads=4)
    want = [int(v) if (k and vv) else None for v, k, vv in zip(oln, ook, valid)]
    pgq.reset_stats()
    ln, ok = st.iterativelength(0, V, ps, pd, src_valid=valid)
    assert lens(ln, ok) == want
    stats = pgq.get_stats()
    assert stats["ball_calls"] >= 1 and stats["ball_segments"] >= len(runs)
    if ball_cap >= 1 << 20 and test_cap >= 1 << 15:
        assert stats["meet_pairs"] >= len(ps) * 0.9  # nearly every row answered by the ball itself
    # (b) sparse directed graph: dead ends, unreachable pairs, long distances -> most rows stay open and come back through the older routes
    V2 = 4000
    st2, ora2 = both(V2, random_graph(rng, V2, 5000), csr_id=1)
    ps, pd, valid = grouped_rows(rng, V2, [40, 1, 300, 1100, 7])
    oln, ook = ora2.lean_iterativelength(V2, ps, pd)
    ln, ok = st2.iterativelength(1, V2, ps, pd)
    assert lens(ln, ok) == [int(v) if k else None for v, k in zip(oln, ook)]
    # (c) a directed ring with chords: distances in the hundreds
    V3 = 3000
    ring = np.arange(V3, dtype=np.int64)
    s3, d3 = np.concatenate([ring, rng.integers(0, V3, 30)]), np.concatenate([(ring + 1) % V3, rng.integers(0, V3, 30)])
    st3, ora3 = both(V3, (s3, d3, np.arange(len(s3), dtype=np.int64)), csr_id=2)
    ps, pd, valid = grouped_rows(rng, V3, [50, 50, 200])
    oln, ook = ora3.lean_iterativelength(V3, ps, pd)
    ln, ok = st3.iterativelength(2, V3, ps, pd)
    assert lens(ln, ok) == [int(v) if k else None for v, k in zip(oln, ook)]
    # (c2) a hub source: over 4096 out-neighbours — its two-hop ball is not walked (S1 only), its rows go on to the other kernels
    V4 = 9000
    hub = np.zeros(6000, dtype=np.int64)
    s4 = np.concatenate([hub, rng.integers(0, V4, 30000)])
    d4 = np.concatenate([rng.choice(np.arange(1, V4), 6000, replace=False), rng.integers(0, V4, 30000)])
    st4, ora4 = both(V4, (s4, d4, np.arange(len(s4), dtype=np.int64)), csr_id=3)
    ps, pd, valid = grouped_rows(rng, V4, [700, 300, 900])
    ps[:700] = 0  # the first run's source is the hub
    oln, ook = ora4.lean_iterativelength(V4, ps, pd)
    ln, ok = st4.iterativelength(3, V4, ps, pd)
    assert lens(ln, ok) == [int(v) if k else None for v, k in zip(oln, ook)]
    # (d) through the bulk entry point on device arrays, 40,000 rows (the decision kernel's range), out-of-range ids refused
    import torch
    ps, pd, _ = grouped_rows(rng, V, [9000, 13000, 1024, 1024, 15952])
    oln, ook = ora.lean_iterativelength(V, ps, pd, nthreads=4)
    dev = st.device_csr(0)
    t_s, t_d = torch.from_numpy(ps).cuda(), torch.from_numpy(pd).cuda()
    t_o = torch.full((len(ps),), -7, dtype=torch.int64, device="cuda")
    dev.iterativelength_bulk_ptr(len(ps), t_s.data_ptr(), t_d.data_ptr(), t_o.data_ptr())
    got = t_o.cpu().numpy()
    assert ((got >= 0) == ook).all() and (got[ook] == oln[ook]).all()
    t_d[5] = V + 3
    with pytest.raises(pgq.PgqError):
        dev.iterativelength_bulk_ptr(len(ps), t_s.data_ptr(), t_d.data_ptr(), t_o.data_ptr())


def test_source_centric_ball_is_chosen_by_the_source_runs():
    # ball = 1 (shipped): k_ball_segments counts the source runs and its last workgroup prices one ball per run + one in-list
    # scan per row against the pre-pass (bytes per row) and the lane batches (level bytes): grouped rows take the ball,
    # scattered pairs the pre-pass, a handful of sources x every vertex the lane batches — same answers on every route.
    rng = np.random.default_rng(61)
    V, E = 20000, 400000
    st, ora = both(V, random_graph(rng, V, E))
    pgq.set_option("meet", 1)
    pgq.set_option("ball", 1)
    pgq.set_option("ball_seg_kb", 16)  # the least a segment costs: 512 KB as shipped — on this 400,000-edge graph a lane batch is cheaper than that
    srcs = rng.choice(V, 40, replace=False)
    for per, expect_ball in ((600, True), (1, False)):
        if per > 1:
            ps = np.repeat(srcs, per)
        else:
            ps = rng.integers(0, V, 24000)
        pd = rng.integers(0, V, len(ps))
        oln, ook = ora.lean_iterativelength(V, ps, pd, nthreads=4)
        for rep in range(2):  # the second call runs under whatever the first one left in the route memo
            pgq.reset_stats()
            ln, ok = st.iterativelength(0, V, ps, pd)
            assert lens(ln, ok) == [int(v) if k else None for v, k in zip(oln, ook)]
            assert (pgq.get_stats()["ball_calls"] >= 1) == expect_ball, (per, rep, pgq.get_stats())
    # the same device buffers, first grouped (the kernel takes them: the next chain on these buffers is its two kernels alone),
    # then overwritten in place with scattered pairs: that chain declines and the stage kernels run after all
    import torch
    dev = st.device_csr(0)
    ps = np.repeat(srcs, 600)
    pd = rng.integers(0, V, len(ps))
    t_s, t_d = torch.from_numpy(ps).cuda(), torch.from_numpy(pd).cuda()
    t_o = torch.empty(len(ps), dtype=torch.int64, device="cuda")
    for rep in range(2):
        dev.iterativelength_bulk_ptr(len(ps), t_s.data_ptr(), t_d.data_ptr(), t_o.data_ptr())
    oln, ook = ora.lean_iterativelength(V, ps, pd, nthreads=4)
    assert (t_o.cpu().numpy() == np.where(ook, oln, -1)).all()
    ps2 = rng.integers(0, V, len(ps))
    t_s.copy_(torch.from_numpy(ps2))
    oln, ook = ora.lean_iterativelength(V, ps2, pd, nthreads=4)
    for rep in range(2):
        pgq.reset_stats()
        dev.iterativelength_bulk_ptr(len(ps), t_s.data_ptr(), t_d.data_ptr(), t_o.data_ptr())
        assert (t_o.cpu().numpy() == np.where(ook, oln, -1)).all()
        assert pgq.get_stats()["ball_calls"] == 0
    # 3 sources x every vertex: one narrow lane batch is cheaper than 60 balls + 60,000 in-list scans
    ps = np.repeat(srcs[:3], V)
    pd = np.tile(np.arange(V, dtype=np.int64), 3)
    oln, ook = ora.lean_iterativelength(V, ps, pd, nthreads=4)
    pgq.reset_stats()
    ln, ok = st.iterativelength(0, V, ps, pd)
    assert (ok == ook).all() and (ln[ok] == oln[ok]).all()


def test_large_grouped_calls_keep_the_route_that_measured_faster():
    # route_timing (shipped on): calls of >= 65,536 rows that the source-centric kernel takes are timed per graph shape; when
    # their best time of at least two calls is over route_try_factor x the lane batches' modelled time, the next two go through
    # the lanes, and from then on through whichever was faster.  Same answers whatever the route; route_try_factor = 0 forces the trial.
    import torch
    rng = np.random.default_rng(71)
    V, E = 20000, 400000
    st, ora = both(V, random_graph(rng, V, E))
    pgq.set_option("meet", 1)
    pgq.set_option("ball", 1)
    pgq.set_option("ball_seg_kb", 16)
    pgq.set_option("calibration_cache", 0)  # (the measured times travel with it: this test wants a graph nobody has timed)
    dev = st.device_csr(0)
    pgq.set_option("route_timing_rows", 16384)
    ps = np.repeat(rng.choice(V, 70, replace=False), 1000)
    pd = rng.integers(0, V, len(ps))
    oln, ook = ora.lean_iterativelength(V, ps, pd, nthreads=4)
    want = np.where(ook, oln, -1)
    t_s, t_d = torch.from_numpy(ps).cuda(), torch.from_numpy(pd).cuda()
    seen = []
    for timing, factor, calls in ((1, 0.0, 6), (0, 0.0, 2)):
        pgq.set_option("route_timing", timing)
        pgq.set_option("route_try_factor", factor)
        for k in range(calls):
            t_o = torch.full((len(ps),), -7, dtype=torch.int64, device="cuda")
            pgq.reset_stats()
            dev.iterativelength_bulk_ptr(len(ps), t_s.data_ptr(), t_d.data_ptr(), t_o.data_ptr())
            assert (t_o.cpu().numpy() == want).all(), (timing, k)
            stt = pgq.get_stats()
            seen.append((timing, k, stt["ball_calls"] >= 1, stt["levels"] > 0))
    assert all(x[2] and not x[3] for x in seen[0:2]), seen  # two calls through the source-centric kernel (a first call pays one-time costs)
    assert all(x[3] and not x[2] for x in seen[2:4]), seen  # two through the lane batches (the trial, best of two)
    assert seen[4][2] != seen[4][3] and seen[5][2:] == seen[4][2:], seen  # then one of the two, and it stays
    assert all(x[2] and not x[3] for x in seen[6:]), seen  # route_timing = 0: the byte models alone
    # what was measured on 70,000 rows says nothing about a call of under half that size (a lane batch costs the same for 32
    # rows per source as for 1000): 30,000 grouped rows go by the byte models, i.e. to the source-centric kernel here
    pgq.set_option("route_timing", 1)
    ps2 = np.repeat(rng.choice(V, 30, replace=False), 1000)
    pd2 = rng.integers(0, V, len(ps2))
    oln2, ook2 = ora.lean_iterativelength(V, ps2, pd2, nthreads=4)
    t_s2, t_d2 = torch.from_numpy(ps2).cuda(), torch.from_numpy(pd2).cuda()
    t_o2 = torch.full((len(ps2),), -7, dtype=torch.int64, device="cuda")
    pgq.reset_stats()
    dev.iterativelength_bulk_ptr(len(ps2), t_s2.data_ptr(), t_d2.data_ptr(), t_o2.data_ptr())
    assert (t_o2.cpu().numpy() == np.where(ook2, oln2, -1)).all()
    assert pgq.get_stats()["ball_calls"] >= 1 and pgq.get_stats()["levels"] == 0


def test_ungrouped_rows_of_few_sources_are_sorted_for_the_source_centric_kernel():
    # ball_sort = 1 (shipped): a cross product whose rows arrive in any order (a hash join's output) is declined by the
    # source-centric kernel (every row a source run of its own) and by the pre-pass (few distinct sources); sorted by source
    # it is the kernel's input after all.  Answers land in the caller's row order; NULL sources sort behind every vertex.
    import torch
    rng = np.random.default_rng(67)
    V, E = 20000, 400000
    st, ora = both(V, random_graph(rng, V, E))
    pgq.set_option("meet", 1)
    pgq.set_option("ball", 1)
    pgq.set_option("ball_seg_kb", 16)
    dev = st.device_csr(0)
    srcs = rng.choice(V, 40, replace=False)
    ps = np.repeat(srcs, 600)
    pd = rng.integers(0, V, len(ps))
    perm = rng.permutation(len(ps))
    ps, pd = ps[perm], pd[perm]
    ps[rng.integers(0, len(ps), 50)] = -1   # NULL rows
    valid = ps >= 0
    oln, ook = ora.lean_iterativelength(V, np.where(valid, ps, 0), np.where(valid, pd, 0), nthreads=4)
    want = np.where(ook & valid, oln, -1)
    t_s, t_d = torch.from_numpy(ps).cuda(), torch.from_numpy(pd).cuda()
    for sort, expect in ((1, True), (0, False)):
        pgq.set_option("ball_sort", sort)
        t_s2, t_d2 = t_s.clone(), t_d.clone()   # fresh buffers: the route memo starts over
        for rep in range(3):  # the second and third call go straight to the sort (route memo)
            t_o = torch.full((len(ps),), -7, dtype=torch.int64, device="cuda")
            pgq.reset_stats()
            dev.iterativelength_bulk_ptr(len(ps), t_s2.data_ptr(), t_d2.data_ptr(), t_o.data_ptr())
            assert (t_o.cpu().numpy() == want).all(), (sort, rep)
            assert (pgq.get_stats()["ball_calls"] >= 1) == expect, (sort, rep, pgq.get_stats())
    pgq.set_option("ball_sort", 1)
    # the same buffers overwritten with scattered pairs: the sorted kernel's rule declines, the memo forgets, same answers
    ps2 = rng.integers(0, V, len(ps))
    t_s2.copy_(torch.from_numpy(ps2))
    oln, ook = ora.lean_iterativelength(V, ps2, pd, nthreads=4)
    for rep in range(2):
        dev.iterativelength_bulk_ptr(len(ps), t_s2.data_ptr(), t_d2.data_ptr(), t_o.data_ptr())
        assert (t_o.cpu().numpy() == np.where(ook, oln, -1)).all()
    # an id outside [0, V) is refused from the sorted route as from every other
    t_s[7] = V + 5
    with pytest.raises(pgq.PgqError):
        dev.iterativelength_bulk_ptr(len(ps), t_s.data_ptr(), t_d.data_ptr(), t_o.data_ptr())


def test_meet_prepass_large_inputs_cross_product_vs_distinct_sources():
    # more than 16384 rows: a sampled estimate of the distinct sources decides between the pre-pass (one two-hop walk per
    # row) and the lane batches (one lane per source)
    rng = np.random.default_rng(12)
    V, E = 20000, 200000
    st, ora = both(V, random_graph(rng, V, E))
    pgq.set_option("meet", 1)
    srcs = rng.integers(0, V, 6)
    ps = np.repeat(srcs, 3500)
    pd = rng.integers(0, V, len(ps))
    oln, ook = ora.lean_iterativelength(V, ps, pd)
    pgq.reset_stats()
    ln, ok = st.iterativelength(0, V, ps, pd)
    assert lens(ln, ok) == [int(v) if k else None for v, k in zip(oln, ook)]
    # cross product: lane batches (only stragglers a batch defers may come back through the pre-pass)
    assert pgq.get_stats()["meet_pairs"] < len(ps) // 4 and pgq.get_stats()["levels"] > 0
    # grouped by source with a group size that a fixed sampling stride would alias with (64 sources x 512 rows: one
    # sampled row per group at a fixed stride would look like distinct pairs); the sampler takes runs of consecutive rows
    srcs = rng.choice(V, 64, replace=False)
    ps = np.repeat(srcs, 512)
    pd = rng.integers(0, V, len(ps))
    oln, ook = ora.lean_iterativelength(V, ps, pd, nthreads=4)
    pgq.reset_stats()
    ln, ok = st.iterativelength(0, V, ps, pd)
    assert lens(ln, ok) == [int(v) if k else None for v, k in zip(oln, ook)]
    assert pgq.get_stats()["meet_pairs"] < len(ps) // 4 and pgq.get_stats()["levels"] > 0
    ps, pd = rng.integers(0, V, 21000), rng.integers(0, V, 21000)
    oln, ook = ora.lean_iterativelength(V, ps, pd)
    pgq.reset_stats()
    ln, ok = st.iterativelength(0, V, ps, pd)
    assert lens(ln, ok) == [int(v) if k else None for v, k in zip(oln, ook)]
    assert pgq.get_stats()["meet_pairs"] > 0  # distinct sources: pre-pass


def test_meet_prepass_out_of_range_ids_rejected():
    import torch
    rng = np.random.default_rng(3)
    V = 500
    s, d, e = random_graph(rng, V, 3000)
    off, adj, eid = graphgen.csr_from_rows(V, s, d)
    dev = pgq.DeviceCSR(V, off, adj, eid)
    pgq.set_option("meet", 1)
    pgq.set_option("meet_bias", 1e9)
    d_src = torch.tensor([1, 2, V + 5], dtype=torch.int64).cuda()
    d_dst = torch.tensor([3, 4, 5], dtype=torch.int64).cuda()
    d_len = torch.empty(3, dtype=torch.int64, device="cuda")
    with pytest.raises(pgq.PgqError):
        dev.iterativelength_bulk_ptr(3, d_src.data_ptr(), d_dst.data_ptr(), d_len.data_ptr())


def test_shared_sources_cross_product(base_config):
    # the binder's shape: few sources x many destinations (match.cpp:467-495) -> lanes are per distinct source
    rng = np.random.default_rng(5)
    V, E = 5000, 40000
    st, ora = both(V, random_graph(rng, V, E))
    srcs = rng.integers(0, V, 7)
    ps = np.repeat(srcs, V)
    pd = np.tile(np.arange(V, dtype=np.int64), 7)
    oln, ook = ora.lean_iterativelength(V, ps, pd, nthreads=4)
    for rep in range(3):  # the second and third call run their levels ahead of the host under the first one's plan
        ln, ok = st.iterativelength(0, V, ps, pd)
        assert (ok == ook).all() and (ln[ok] == oln[ok]).all()
    assert pgq.get_stats()["unique_sources"] >= 1


def test_literal_reference_restatement_agrees():
    # the literal 512-lane restatement (what cpu_baseline times) on a mid-size graph, > 512 pairs
    rng = np.random.default_rng(9)
    V, E = 20000, 120000
    st, ora = both(V, random_graph(rng, V, E))
    n = 1200
    ps, pd = rng.integers(0, V, n), rng.integers(0, V, n)
    ln, ok = st.iterativelength(0, V, ps, pd)
    oln, ook = ora.iterativelength(V, ps, pd)
    assert lens(ln, ok) == lens(oln, ook)
    assert st.shortestpath(0, V, ps[:600], pd[:600]) == ora.shortestpath(V, ps[:600], pd[:600])


def test_rmat18_and_snb_like_medium():
    V, s, d = graphgen.rmat(18, seed=18)
    off, adj, eid = graphgen.csr_from_rows(V, s, d)
    ora = OracleCSR.adopt(V, off, adj, eid)
    dev = pgq.DeviceCSR(V, off, adj, eid)
    rng = np.random.default_rng(2)
    n = 4096
    ps, pd = rng.integers(0, V, n), rng.integers(0, V, n)
    oln, ook = ora.lean_iterativelength(V, ps, pd, nthreads=8)
    for words in (0, 4, 32):
        pgq.set_option("words", words)
        ln, ok = dev.iterativelength(ps, pd)
        assert (ok == ook).all() and (ln[ok] == oln[ok]).all()
    paths = dev.shortestpath(ps[:1024], pd[:1024])
    assert paths == ora.lean_shortestpath(V, ps[:1024], pd[:1024])
    pgq.set_option("words", 0)
    V2, s2, d2 = graphgen.snb_knows_like(V=20000, friendships=600_000, seed=3)
    off2, adj2, eid2 = graphgen.csr_from_rows(V2, s2, d2)
    ora2 = OracleCSR.adopt(V2, off2, adj2, eid2)
    dev2 = pgq.DeviceCSR(V2, off2, adj2, eid2)
    ps, pd = rng.integers(0, V2, n), rng.integers(0, V2, n)
    ln, ok = dev2.iterativelength(ps, pd)
    oln, ook = ora2.lean_iterativelength(V2, ps, pd, nthreads=8)
    assert (ok == ook).all() and (ln[ok] == oln[ok]).all()
    # undirected graph: d(s,t) == d(t,s) (size-independent property)
    ln_r, ok_r = dev2.iterativelength(pd, ps)
    assert (ok_r == ok).all() and (ln_r == ln).all()
    paths = dev2.shortestpath(ps[:1024], pd[:1024])
    assert paths == ora2.lean_shortestpath(V2, ps[:1024], pd[:1024])


# ---- cheapest path ---------------------------------------------------------------------------------------------

@pytest.mark.parametrize("kind", ["int64", "double"])
def test_cheapest_path_bit_exact(kind):
    rng = np.random.default_rng(13)
    V, E = 4000, 30000
    s, d, e = random_graph(rng, V, E, skew=True)
    w = rng.integers(1, 1000, E) if kind == "int64" else rng.random(E) + 0.01
    st, ora = both(V, (s, d, e), w=w)
    for n, div, light, split, streams in (
            (1, 64, 0, 1, 2), (70, 64, 0, 1, 2), (300, 64, 0, 1, 2), (300, 0, 0, 1, 2), (300, 1, 0, 0, 2), (300, 100000, 0, 1, 2),
            (1500, 64, 0, 1, 2), (1500, 4, 0, 0, 1), (1500, 0, 0, 1, 4),
            (1, 0, 1, 1, 2), (300, 0, 1, 1, 2), (300, 0, 4, 0, 2), (1500, 0, 1, 1, 4), (1500, 0, 4, 1, 1), (1500, 4, 1000, 1, 3)):
        # lists longer than 128 edges (the hubs of the skewed graph) relaxed 64 edges per wavefront by a second launch, or not
        pgq.set_option("relax_split", split)
        pgq.set_option("streams", streams)  # batches of 64 sources side by side on their own label arrays
        pgq.set_option("relax_small_limit", 0 if n == 300 else (50 if n == 70 else 2048))  # host rounds / mixed / device rounds
        # light edges first: weight-sorted lists under a cap that doubles per phase, first cap = mean weight / light (0: off)
        pgq.set_option("relax_light", 2 if light else 0)  # (1 would go by the mean out-degree: 7.5 here, plain rounds)
        pgq.set_option("relax_light_div", max(light, 1))
        pgq.set_option("relax_labels32", (n + streams) % 2)  # int64 weights under the light path: 4-byte / 8-byte labels
        # band width of the ordered rounds = mean weight / div: 0 = plain rounds, 1 = a few wide bands, 100000 = a band per label
        pgq.set_option("relax_delta_div", div)
        ps, pd = rng.integers(0, V, n), rng.integers(0, V, n)
        if div == 4:  # many rows per lane: the per-lane bound is the largest of its destinations' labels
            ps = ps[rng.integers(0, 40, n)]
        out, ok = st.cheapest_path_length(0, V, ps, pd)
        lout, lok = ora.lean_cheapest_path_length(V, ps, pd)
        assert (ok == lok).all()
        assert (out[ok] == lout[ok]).all()  # bit-exact, doubles included
        assert out.dtype == lout.dtype
    # literal Bellman-Ford restatement (cheapest_path_length.cpp:52-136) on a batch that hits the 256-lane path
    ps, pd = rng.integers(0, V, 300), rng.integers(0, V, 300)
    out, ok = st.cheapest_path_length(0, V, ps, pd)
    rout, rok = ora.cheapest_path_length(V, ps, pd)
    assert (ok == rok).all() and (out[ok] == rout[ok]).all()
    # NULL dst -> NULL (cheapest_path_length.cpp:74-76)
    dv = np.ones(300, dtype=bool)
    dv[::7] = False
    out, ok = st.cheapest_path_length(0, V, ps, pd, dst_valid=dv)
    assert (ok == (rok & dv)).all()


@pytest.mark.parametrize("case", ["signed_zero", "inf_nan", "wide_int64", "chain_inf_nan"])
def test_cheapest_path_special_weights(case):
    """Weights the weight-sorted relaxation must not trip over: -0.0 (accepted — it is not < 0 — and its bit pattern would
    sort last), +inf / NaN (never relax an edge; no usable mean, so plain rounds), int64 weights over 40 binary orders
    (many cap doublings).  Against the literal Bellman-Ford restatement (cheapest_path_length.cpp:52-136)."""
    rng = np.random.default_rng(77)
    V, E = 2000, 14000
    s, d, e = random_graph(rng, V, E, skew=True)
    if case == "chain_inf_nan":  # a forest (one out-edge per vertex): rows answered by the chain walk, which must apply
        # the reference's test against the max/2 sentinel — an inf / NaN / 1e308 weight on the way means NULL, not a sum
        E = V - 1
        s = np.arange(1, V, dtype=np.int64)
        d = (rng.random(E) * s).astype(np.int64)
        e = np.arange(E, dtype=np.int64)
    if case == "wide_int64":
        w = (2.0 ** (rng.random(E) * 40)).astype(np.int64)
    else:
        w = rng.random(E) + 0.01
        w[rng.random(E) < 0.06] = -0.0
        w[rng.random(E) < 0.06] = 0.0
        if case in ("inf_nan", "chain_inf_nan"):
            w[rng.random(E) < 0.02] = np.inf
            w[rng.random(E) < 0.02] = np.nan
            w[rng.random(E) < 0.02] = np.copysign(np.nan, -1.0)  # sign bit set: the bit pattern is "smaller" than any label
            w[rng.random(E) < 0.01] = 1e308                        # a finite sum that does not get under max/2
    st, ora = both(V, (s, d, e), w=w)
    for light, streams in ((1, 1), (1, 3), (0, 2)):
        pgq.set_option("relax_light", light)
        pgq.set_option("streams", streams)
        pgq.set_option("relax_small_limit", 0 if light else 2048)
        ps, pd = rng.integers(0, V, 300), rng.integers(0, V, 300)
        if case == "chain_inf_nan":  # destinations a few hops up the chain
            pd = ps.copy()
            for _ in range(int(rng.integers(1, 6))):
                up = pd > 0
                pd[up] = d[pd[up] - 1]
        out, ok = st.cheapest_path_length(0, V, ps, pd)
        rout, rok = ora.cheapest_path_length(V, ps, pd)
        assert (ok == rok).all()
        assert out[ok].tobytes() == rout[ok].tobytes()  # bit for bit (a -0.0 label would differ from +0.0 here)


@pytest.mark.parametrize("delta_div", [8, 1, 100000])
def test_weighted_pair_search_bit_exact(delta_div):
    # k_wbibfs (pgq_cheapest.hip): bidirectional band-wise label correcting per row, int64 weights; delta_div sets the
    # band width (mean weight / delta_div): wide bands (1), the default, and one-unit bands (100000)
    rng = np.random.default_rng(500 + delta_div % 97)
    pgq.set_option("wbibfs", 1)
    pgq.set_option("wbibfs_delta_div", delta_div)
    cases = []
    V, E = 4000, 30000
    s, d, e = random_graph(rng, V, E, skew=True)
    cases.append((V, (s, d, e), rng.integers(1, 1000, E)))                      # skewed, hubs
    cases.append((V, (s, d, e), rng.integers(0, 3, E)))                         # many zero-weight edges and ties
    s2, d2, e2 = random_graph(rng, 6000, 7000)
    cases.append((6000, (s2, d2, e2), rng.integers(1, 50, 7000)))               # sparse: unreachable pairs, small closures
    s3 = np.concatenate([s[:5000], s[:5000]])
    d3 = np.concatenate([d[:5000], d[:5000]])
    cases.append((V, (s3, d3, np.arange(10000, dtype=np.int64)), rng.integers(1, 9, 10000)))  # parallel edges
    for cid, (Vc, rows, w) in enumerate(cases):
        st, ora = both(Vc, rows, w=w.astype(np.int64), csr_id=cid)
        n = 700
        ps, pd = rng.integers(0, Vc, n), rng.integers(0, Vc, n)
        ps[:20] = pd[:20]
        pgq.reset_stats()
        out, ok = st.cheapest_path_length(cid, Vc, ps, pd)
        lout, lok = ora.lean_cheapest_path_length(Vc, ps, pd)
        assert (ok == lok).all() and (out[ok] == lout[ok]).all()
        assert pgq.get_stats()["meet_pairs"] > 0
    # caps: rows over the work / queue caps are left to the batched relaxation (mixed answers)
    st, ora = both(cases[0][0], cases[0][1], w=cases[0][2].astype(np.int64), csr_id=9)
    ps, pd = rng.integers(0, V, 300), rng.integers(0, V, 300)
    lout, lok = ora.lean_cheapest_path_length(V, ps, pd)
    for cap, queue, far in ((2000, 1 << 17, 1 << 21), (64 << 20, 1024, 1 << 21), (64 << 20, 1 << 17, 1024)):
        pgq.set_option("wbibfs_cap", cap)
        pgq.set_option("wbibfs_queue", queue)
        pgq.set_option("wbibfs_far", far)
        out, ok = st.cheapest_path_length(9, V, ps, pd)
        assert (ok == lok).all() and (out[ok] == lout[ok]).all()
    pgq.set_option("wbibfs_far", 1 << 21)
    # double weights keep the batched relaxation (the two-sided sum is not the reference's left fold)
    pgq.set_option("wbibfs_cap", 64 << 20)
    pgq.set_option("wbibfs_queue", 1 << 17)
    st, ora = both(V, cases[0][1], w=rng.random(E) + 0.01, csr_id=10)
    out, ok = st.cheapest_path_length(10, V, ps[:100], pd[:100])
    lout, lok = ora.lean_cheapest_path_length(V, ps[:100], pd[:100])
    assert (ok == lok).all() and (out[ok] == lout[ok]).all()


@pytest.mark.parametrize("labels32", [1, 0])
def test_two_ended_relaxation_bit_exact(labels32):
    # relax_batches_bidir (pgq_cheapest.hip): every lane a (src, dst) pair, k_relax from both ends under a common distance cap;
    # int64 weights, lists of pairs.  Caps: shipped, one-unit first cap / steps (a phase per label), one huge cap (a single phase)
    rng = np.random.default_rng(811 + labels32)
    pgq.set_option("relax_bidir", 1)
    pgq.set_option("relax_light", 2)
    pgq.set_option("relax_labels32", labels32)
    pgq.set_option("chain", 0)
    cases = []
    V, E = 4000, 30000
    s, d, e = random_graph(rng, V, E, skew=True)
    cases.append((V, (s, d, e), rng.integers(1, 1000, E)))                      # skewed, hubs over 128 edges (heavy lists)
    cases.append((V, (s, d, e), rng.integers(0, 3, E)))                         # many zero-weight edges and ties
    s2, d2, e2 = random_graph(rng, 6000, 7000)
    cases.append((6000, (s2, d2, e2), rng.integers(1, 50, 7000)))               # sparse: unreachable pairs, small closures
    s3 = np.concatenate([s[:5000], s[:5000]])
    d3 = np.concatenate([d[:5000], d[:5000]])
    cases.append((V, (s3, d3, np.arange(10000, dtype=np.int64)), rng.integers(1, 9, 10000)))  # parallel edges
    cases.append((V, (s, d, e), np.full(E, 7)))                                 # one weight: every band but each seventh is empty
    cases.append((V, (s, d, e), (2.0 ** (rng.random(E) * 30)).astype(np.int64)))  # thirty binary orders (8-byte labels)
    for cid, (Vc, rows, w) in enumerate(cases):
        st, ora = both(Vc, rows, w=w.astype(np.int64), csr_id=cid)
        for n, c0, step, streams in ((700, 16, 32, 2), (64, 16, 32, 1), (1, 16, 32, 2), (130, 1 << 30, 1 << 30, 3), (200, 1, 1, 2)):
            pgq.set_option("relax_bidir_c0_div", c0)
            pgq.set_option("relax_bidir_step_div", step)
            pgq.set_option("streams", streams)
            ps, pd = rng.integers(0, Vc, n), rng.integers(0, Vc, n)
            if n >= 64:
                ps[:20] = pd[:20]          # trivial rows
                ps[20:30] = ps[30]         # a source shared by several lanes (still <= 2 rows per distinct source overall)
                pd[40:48] = pd[48]         # a destination shared by several lanes
            out, ok = st.cheapest_path_length(cid, Vc, ps, pd)
            lout, lok = ora.lean_cheapest_path_length(Vc, ps, pd)
            assert (ok == lok).all() and (out[ok] == lout[ok]).all(), (cid, n, c0)
    # the two-ended search is taken (fewer relaxed edges than one lane per source), a cross product is not (same as without it)
    st, ora = both(cases[0][0], cases[0][1], w=cases[0][2].astype(np.int64), csr_id=9)
    ps, pd = rng.integers(0, V, 400), rng.integers(0, V, 400)
    lout, lok = ora.lean_cheapest_path_length(V, ps, pd)
    scanned = {}
    for bidir in (1, 0):
        pgq.set_option("relax_bidir", bidir)
        pgq.reset_stats()
        out, ok = st.cheapest_path_length(9, V, ps, pd)
        assert (ok == lok).all() and (out[ok] == lout[ok]).all()
        scanned[bidir] = pgq.get_stats()["edges_scanned"]
    assert scanned[1] != scanned[0]
    ps = ps[rng.integers(0, 20, 400)]  # 20 sources x 20 destinations each
    lout, lok = ora.lean_cheapest_path_length(V, ps, pd)
    for bidir in (1, 0):
        pgq.set_option("relax_bidir", bidir)
        pgq.reset_stats()
        out, ok = st.cheapest_path_length(9, V, ps, pd)
        assert (ok == lok).all() and (out[ok] == lout[ok]).all()
        scanned[bidir] = pgq.get_stats()["batches"]
    assert scanned[1] == scanned[0] == 1  # 20 sources: one batch of lanes either way (400 pairs would be 7 batches of pairs)
    # double weights keep one lane per source (a two-sided sum is not the reference's left fold)
    pgq.set_option("relax_bidir", 1)
    st, ora = both(V, cases[0][1], w=rng.random(E) + 0.01, csr_id=10)
    ps, pd = rng.integers(0, V, 100), rng.integers(0, V, 100)
    out, ok = st.cheapest_path_length(10, V, ps, pd)
    lout, lok = ora.lean_cheapest_path_length(V, ps, pd)
    assert (ok == lok).all() and (out[ok] == lout[ok]).all()


def test_cheapest_forest_and_zero_weights():
    rep = load_golden("snb003_replyof.json")
    e = np.asarray(rep["edges"], dtype=np.int64)
    rng = np.random.default_rng(3)
    w = rng.integers(0, 50, len(e))  # zeros allowed
    st, ora = both(rep["V"], (e[:, 0], e[:, 1], np.arange(len(e), dtype=np.int64)), w=w)
    ps = rng.integers(0, rep["V"], 900)
    pd = rng.integers(0, rep["V"], 900)
    ps[:400], pd[:400] = e[:400, 0], e[:400, 1]
    out, ok = st.cheapest_path_length(0, rep["V"], ps, pd)
    lout, lok = ora.lean_cheapest_path_length(rep["V"], ps, pd)
    assert (ok == lok).all() and (out[ok] == lout[ok]).all()


def test_errors_from_the_device_layer():
    g = load_golden("student_directed.json")
    st, _ = both(g["V"], directed_rows(g["edges"]))
    with pytest.raises(pgq.PgqError, match="out of range"):
        st.iterativelength(0, g["V"], [0], [99])
    with pytest.raises(pgq.PgqError, match="Need to initialize CSR before doing cheapest path"):
        st.cheapest_path_length(0, g["V"], [0], [1])


def test_traversed_edges_accounting_matches_oracle():
    # the MTEPS numerator bench.py reports: per-pair traversed edges, GPU accounting vs the oracle's own BFS
    import torch
    rng = np.random.default_rng(21)
    V, E = 6000, 50000
    s, d, e = random_graph(rng, V, E, skew=True)
    off, adj, eid = graphgen.csr_from_rows(V, s, d)
    ora = OracleCSR.adopt(V, off, adj, eid)
    dev = pgq.DeviceCSR(V, off, adj, eid)
    n = 3000
    ps, pd = rng.integers(0, V, n), rng.integers(0, V, n)
    oln, ook, ote = ora.lean_iterativelength(V, ps, pd, with_te=True)
    for words in (2, 32):
        pgq.set_option("words", words)
        d_src, d_dst = torch.from_numpy(ps).cuda(), torch.from_numpy(pd).cuda()
        d_len = torch.empty(n, dtype=torch.int64, device="cuda")
        d_te = torch.empty(n, dtype=torch.int64, device="cuda")
        dev.traversed_edges_bulk_ptr(n, d_src.data_ptr(), d_dst.data_ptr(), d_len.data_ptr(), d_te.data_ptr())
        ln = d_len.cpu().numpy()
        assert ((ln >= 0) == ook).all() and (ln[ook] == oln[ook]).all()
        assert (d_te.cpu().numpy() == ote).all()


@pytest.mark.slow
def test_full_size_sf100_properties():
    """BASELINE-size graph (SF100-shaped knows, 39.9 M CSR entries): properties that need no oracle run.
    undirected symmetry d(s,t) == d(t,s); every reported path is a real path of the reported length whose
    vertices sit at consecutive BFS distances; a bounded oracle sample agrees bit for bit."""
    V, s, d = graphgen.snb_knows_like()
    off, adj, eid = graphgen.csr_from_rows(V, s, d)
    dev = pgq.DeviceCSR(V, off, adj, eid)
    rng = np.random.default_rng(3)
    n = 4096
    ps, pd = rng.integers(0, V, n), rng.integers(0, V, n)
    ln, ok = dev.iterativelength(ps, pd)
    ln_r, ok_r = dev.iterativelength(pd, ps)
    assert (ok == ok_r).all() and (ln == ln_r).all()
    paths = dev.shortestpath(ps, pd)
    for i in range(n):
        p = paths[i]
        if not ok[i]:
            assert p is None
            continue
        assert p is not None and len(p) == 2 * ln[i] + 1 and p[0] == ps[i] and p[-1] == pd[i]
        for k in range(0, len(p) - 1, 2):  # edge id = CSR slot's edge rowid: slot must run from p[k] to p[k+2]
            u, e, v = p[k], p[k + 1], p[k + 2]
            row = adj[off[u]:off[u + 1]]
            j = int(np.argmax(row == v))
            assert row[j] == v and eid[off[u] + j] == e  # first slot of u holding v (shortest_path.cpp:23-30)
    ora = OracleCSR.adopt(V, off, adj, eid)
    oln, ook = ora.lean_iterativelength(V, ps, pd, nthreads=16)
    assert (ook == ok).all() and (oln[ook] == ln[ook]).all()
    # 1024 full paths against the oracle: the min-id parent / first-slot tie-break of shortest_path.cpp:21-31
    assert ora.lean_shortestpath(V, ps[:1024], pd[:1024]) == paths[:1024]
    pgq.set_option("meet", 1)  # shortestpath through the pre-pass: same lists, BASELINE configs[2] size
    assert dev.shortestpath(ps, pd) == paths
    pgq.set_option("meet", 0)
    # the pair-centric pre-pass (k_meet3 / k_meet4) against the lane-batched search and the oracle, BASELINE configs[3]
    # size: 65,536 pairs
    import torch
    allp = np.random.default_rng(4).integers(0, V, size=(65536, 2))
    d_src, d_dst = torch.from_numpy(allp[:, 0].copy()).cuda(), torch.from_numpy(allp[:, 1].copy()).cuda()
    d_a = torch.empty(65536, dtype=torch.int64, device="cuda")
    d_b = torch.empty(65536, dtype=torch.int64, device="cuda")
    pgq.set_option("meet", 1)
    pgq.reset_stats()
    dev.iterativelength_bulk_ptr(65536, d_src.data_ptr(), d_dst.data_ptr(), d_a.data_ptr())
    assert pgq.get_stats()["meet_pairs"] > 60000
    pgq.set_option("meet", 0)
    dev.iterativelength_bulk_ptr(65536, d_src.data_ptr(), d_dst.data_ptr(), d_b.data_ptr())
    assert bool((d_a == d_b).all())
    oln, ook = ora.lean_iterativelength(V, allp[:8192, 0], allp[:8192, 1], nthreads=16)
    got = d_a[:8192].cpu().numpy()
    assert ((got >= 0) == ook).all() and (got[ook] == oln[ook]).all()


@pytest.mark.slow
def test_c2_rmat22_1024_pairs_matches_literal_oracle():
    """BASELINE configs[1] at its stated scale: R-MAT scale 22 (4.2 M vertices, 67 M directed edges), 1024 pairs of
    default_rng(2), iterativelength — lane-batched search and pre-pass against the literal 512-lane restatement."""
    V, s, d = graphgen.rmat(22, seed=22)
    off, adj, eid = graphgen.csr_from_rows(V, s, d)
    dev = pgq.DeviceCSR(V, off, adj, eid)
    pairs = np.random.default_rng(2).integers(0, V, size=(1024, 2))
    ora = OracleCSR.adopt(V, off, adj, eid)
    oln, ook = ora.baseline_run("iterativelength", V, pairs[:, 0], pairs[:, 1], nthreads=2)
    for meet in (0, 1):
        pgq.set_option("meet", meet)
        pgq.set_option("meet_bias", 1e9)
        ln, ok = dev.iterativelength(pairs[:, 0], pairs[:, 1])
        assert (ok == ook).all() and (ln[ok] == oln[ook]).all()


def test_c5_weighted_cheapest_path_at_scale_int64_and_double():
    """BASELINE configs[4] shape: 4096 pairs with REACHABLE destinations (ancestors of the source) on a 2^20-vertex
    reply forest, and 1024 pairs on a weighted SNB-like graph; int64 and double weights, against Dijkstra."""
    rng = np.random.default_rng(55)
    V, s, d = graphgen.reply_forest(1 << 20, seed=5)
    off, adj, eid = graphgen.csr_from_rows(V, s, d)
    deg = np.diff(off)
    cand = np.nonzero(deg > 0)[0]
    src = cand[rng.integers(0, len(cand), 4096)]
    dst = src.copy()
    hops = rng.integers(1, 9, 4096)
    for h in range(8):
        move = (hops > h) & (deg[dst] > 0)
        dst[move] = adj[off[dst[move]]]
    dst[:256] = rng.integers(0, V, 256)  # some unreachable ones
    for w in (rng.integers(1, 1000, len(adj)), rng.random(len(adj)) * 10.0):
        dev = pgq.DeviceCSR(V, off, adj, eid, w)
        ora = OracleCSR.adopt(V, off, adj, eid, w)
        out, ok = dev.cheapest_path_length(src, dst)
        want, wok = ora.lean_cheapest_path_length(V, src, dst)
        assert (ok == wok).all() and (out[ok] == want[wok]).all() and ok.sum() > 3000
    V2, s2, d2 = graphgen.snb_knows_like(20000, 400000, seed=9)
    off2, adj2, eid2 = graphgen.csr_from_rows(V2, s2, d2)
    ps, pd = rng.integers(0, V2, 1024), rng.integers(0, V2, 1024)
    for w in (rng.integers(1, 100, len(adj2)), rng.random(len(adj2)) + 0.01):
        dev = pgq.DeviceCSR(V2, off2, adj2, eid2, w)
        ora = OracleCSR.adopt(V2, off2, adj2, eid2, w)
        out, ok = dev.cheapest_path_length(ps, pd)
        want, wok = ora.lean_cheapest_path_length(V2, ps, pd)
        assert (ok == wok).all() and (out[ok] == want[wok]).all()


def _dijkstra_threads(ora, V, ps, pd, threads=16):
    """the oracle's per-pair Dijkstra on slices of the rows, one slice per host thread (the call releases the GIL and
    the oracle CSR is read-only)"""
    from concurrent.futures import ThreadPoolExecutor
    parts = np.array_split(np.arange(len(ps)), threads)
    with ThreadPoolExecutor(threads) as ex:
        res = list(ex.map(lambda ix: ora.lean_cheapest_path_length(V, ps[ix], pd[ix]), parts))
    return np.concatenate([r[0] for r in res]), np.concatenate([r[1] for r in res])


@pytest.mark.slow
def test_sf100_cross_product_2048x1024_and_weighted_pairs_against_the_oracle():
    """The bench's other two legs at their full size, against the oracle (round-3 review: the bench compared the first
    8192 rows of the cross product = 8 of its 2048 sources, and 32 weighted rows; no test ran either shape at SF100 scale).
    (1) 2048 distinct sources x 1024 destinations = 2.1 M rows through the product's own routing (lane-batched MS-BFS,
    k_pull<32> at 32 lane-words on the real graph): a strided sample of 8192 rows that touches every source, against the
    oracle's per-pair BFS.  (2) 512 random pairs on the same graph with int64 weights 1..999 and with double weights,
    cheapest_path_length against the oracle's Dijkstra, bit for bit."""
    import torch
    for k, v in (("meet", 1), ("meet_bias", 1.0), ("streams", 3), ("probe2_abs", 4096), ("push_div", 24)):
        pgq.set_option(k, v)  # the shipped values
    V, s, d = graphgen.snb_knows_like()
    off, adj, eid = graphgen.csr_from_rows(V, s, d)
    dev = pgq.DeviceCSR(V, off, adj, eid)
    ora = OracleCSR.adopt(V, off, adj, eid)
    rng = np.random.default_rng(7)
    src = rng.choice(V, size=2048, replace=False)
    ps = np.repeat(src, 1024)
    pd = rng.integers(0, V, len(ps))
    d_src, d_dst = torch.from_numpy(ps).cuda(), torch.from_numpy(pd).cuda()
    d_out = torch.full((len(ps),), -7, dtype=torch.int64, device="cuda")
    pgq.reset_stats()
    dev.iterativelength_bulk_ptr(len(ps), d_src.data_ptr(), d_dst.data_ptr(), d_out.data_ptr())
    st = pgq.get_stats()
    assert st["levels"] > 0 and st["unique_sources"] == 2048  # the lane-batched search ran, one lane per source
    sel = np.arange(8192) * (len(ps) // 8192)
    assert len(np.unique(ps[sel])) == 2048
    got = d_out.cpu().numpy()
    assert (got != -7).all()
    oln, ook = ora.lean_iterativelength(V, ps[sel], pd[sel], nthreads=16)
    assert ((got[sel] >= 0) == ook).all() and (got[sel][ook] == oln[ook]).all()
    dev.close()
    wp = np.random.default_rng(106).integers(0, V, size=(512, 2))
    wrng = np.random.default_rng(6)
    for w in (wrng.integers(1, 1000, len(adj)), wrng.integers(1, 1000, len(adj)).astype(np.float64) / 7.0):
        devw = pgq.DeviceCSR(V, off, adj, eid, w)
        out, ok = devw.cheapest_path_length(wp[:, 0], wp[:, 1])
        want, wok = _dijkstra_threads(OracleCSR.adopt(V, off, adj, eid, w), V, wp[:, 0], wp[:, 1])
        assert (ok == wok).all() and (out[ok] == want[wok]).all() and ok.sum() > 500
        devw.close()


def test_upload_reads_only_the_first_v_V_entries():
    """The undirected CTE allocates e / edge_ids / w at twice the size the CSR uses (compressed_sparse_row.cpp:127,166: both
    size arguments are the doubled count; only e[0 .. v[V]) is meaningful, the tail is whatever the allocation held).
    Host arrays with a junk tail past v[V] — ids outside [0, V), huge weights — must give the same device CSR."""
    V, s, d = graphgen.rmat(10, seed=3)
    off, adj, eid = graphgen.csr_from_rows(V, s, d)
    E = int(off[V])
    rng = np.random.default_rng(1)
    w = rng.integers(1, 50, E)
    junk = np.full(E + 17, 1 << 40, dtype=np.int64)
    adj_j, eid_j, w_j = (np.concatenate([x, junk]) for x in (adj, eid, w))
    adj_j[E:] = -5  # not a vertex
    off_j = np.concatenate([off, [E]])  # the reference allocates V + 2 offsets
    ps, pd = rng.integers(0, V, 600), rng.integers(0, V, 600)
    ora = OracleCSR.adopt(V, off, adj, eid, w)
    for meet in (0, 1):
        pgq.set_option("meet", meet)
        pgq.set_option("meet_bias", 1e9)
        dev = pgq.DeviceCSR(V, off_j, adj_j, eid_j, w_j)
        ln, ok = dev.iterativelength(ps, pd)
        oln, ook = ora.lean_iterativelength(V, ps, pd)
        assert (ok == ook).all() and (ln[ok] == oln[ook]).all()
        assert dev.shortestpath(ps[:200], pd[:200]) == ora.lean_shortestpath(V, ps[:200], pd[:200])
        out, cok = dev.cheapest_path_length(ps[:200], pd[:200])
        want, wok = ora.lean_cheapest_path_length(V, ps[:200], pd[:200])
        assert (cok == wok).all() and (out[cok] == want[wok]).all()
        dev.close()


def test_bulk_device_entry_points_match_chunk_api():
    """pgq_shortestpath_bulk_device / pgq_cheapest_path_length_bulk_device (external child buffer, overflow return,
    straggler append) against the chunk API and the oracle."""
    import torch
    rng = np.random.default_rng(8)
    V, E = 30000, 400000
    s, d, e = random_graph(rng, V, E, skew=True)
    off, adj, eid = graphgen.csr_from_rows(V, s, d)
    w = rng.integers(1, 50, E)[eid]
    dev = pgq.DeviceCSR(V, off, adj, eid, w)
    ora = OracleCSR.adopt(V, off, adj, eid, w)
    n = 6000
    ps, pd = rng.integers(0, V, n), rng.integers(0, V, n)
    ps[:20] = pd[:20]
    want = ora.lean_shortestpath(V, ps, pd)
    d_src, d_dst = torch.from_numpy(ps).cuda(), torch.from_numpy(pd).cuda()
    d_len = torch.empty(n, dtype=torch.int64, device="cuda")
    d_off = torch.zeros(n, dtype=torch.int64, device="cuda")
    need = sum(len(p) for p in want if p is not None)
    # defer 2: stragglers of wide batches are appended by a narrow second pass; meet: answered rows first, rest appended
    for words, defer, meet in ((0, 8, 0), (8, 2, 0), (0, 8, 1)):
        pgq.set_option("words", words)
        pgq.set_option("defer", defer)
        pgq.set_option("meet", meet)
        pgq.set_option("meet_bias", 1e9)
        pgq.set_option("meet_cap", 20000)
        d_child = torch.empty(need + 16, dtype=torch.int64, device="cuda")
        rc, used = dev.shortestpath_bulk_ptr(n, d_src.data_ptr(), d_dst.data_ptr(), d_len.data_ptr(), d_off.data_ptr(),
                                             d_child.data_ptr(), need + 16)
        assert rc == 0 and used == need
        ln, of, ch = d_len.cpu().numpy(), d_off.cpu().numpy(), d_child.cpu().numpy()
        got = [None if ln[i] < 0 else ch[of[i]:of[i] + 2 * ln[i] + 1].tolist() for i in range(n)]
        assert got == want
        assert got == dev.shortestpath(ps, pd)
        # too small a child buffer: error status, `used` reports what is needed, lengths are still right
        small = torch.empty(need // 2, dtype=torch.int64, device="cuda")
        rc, used = dev.shortestpath_bulk_ptr(n, d_src.data_ptr(), d_dst.data_ptr(), d_len.data_ptr(), d_off.data_ptr(),
                                             small.data_ptr(), need // 2)
        assert rc != 0 and used == need
        assert (d_len.cpu().numpy() == ln).all()
    pgq.set_option("words", 0)
    pgq.set_option("defer", 8)
    pgq.set_option("meet", 0)
    d_val = torch.zeros(n, dtype=torch.int64, device="cuda")
    d_ok = torch.zeros(n, dtype=torch.uint8, device="cuda")
    dev.cheapest_bulk_ptr(n, d_src.data_ptr(), d_dst.data_ptr(), d_val.data_ptr(), d_ok.data_ptr())
    cw, cok = ora.lean_cheapest_path_length(V, ps, pd)
    ok = d_ok.cpu().numpy().astype(bool)
    assert (ok == cok).all() and (d_val.cpu().numpy()[ok] == cw[cok]).all()
    out, ok2 = dev.cheapest_path_length(ps, pd)
    assert (ok2 == cok).all() and (out[ok2] == cw[cok]).all()


def test_device_csr_construction_matches_reference_layout():
    # create_csr_vertex/create_csr_edge on the GPU: same v/e/edge_ids/w arrays as the
    # single-threaded reference schedule, golden layout of getpgschema.test:85-107 included
    import torch
    g = load_golden("student_csr_layout.json")
    s, d, e = directed_rows(g["edges"])
    ts, td = torch.from_numpy(s).cuda(), torch.from_numpy(d).cuda()  # keep the tensors alive across the call
    dev = pgq.DeviceCSR.build_from_device_rows(g["V"], len(s), ts.data_ptr(), td.data_ptr())
    off, adj, eid, _ = dev.download()
    assert off.tolist() == g["csr_v"][:g["V"] + 1] and adj.tolist() == g["csr_e"]
    rng = np.random.default_rng(17)
    V, E = 5000, 60000
    s, d, e = random_graph(rng, V, E, skew=True)
    e = rng.permutation(E).astype(np.int64) + 1000  # arbitrary edge rowids
    for w in (None, rng.integers(0, 1000, E), rng.random(E)):
        ts, td, te = (torch.from_numpy(x).cuda() for x in (s, d, e))
        tw = None if w is None else torch.from_numpy(w).cuda()
        dev = pgq.DeviceCSR.build_from_device_rows(V, E, ts.data_ptr(), td.data_ptr(), te.data_ptr(),
                                                   0 if tw is None else tw.data_ptr(),
                                                   0 if w is None else (2 if w.dtype.kind == "f" else 1))
        ora = OracleCSR.from_edges(V, s, d, e, w)
        off, adj, eid, ww = dev.download()
        assert (off == ora.v[:V + 1]).all() and (adj == ora.e).all() and (eid == ora.edge_ids).all()
        if w is not None:
            assert (ww == ora.w).all()
        ps, pd = rng.integers(0, V, 500), rng.integers(0, V, 500)
        assert dev.shortestpath(ps, pd) == ora.lean_shortestpath(V, ps, pd)
    with pytest.raises(pgq.PgqError, match="out of range"):
        bad = torch.from_numpy(np.array([0, V], dtype=np.int64)).cuda()
        pgq.DeviceCSR.build_from_device_rows(V, 2, bad.data_ptr(), bad.data_ptr())


def test_fuzz_tiny_graphs_against_literal_oracle(product_config):
    """Many tiny graphs (self loops, duplicate and anti-parallel edges, isolated vertices, V=1, E=0, chains) through
    the UDF mirror, every function against the literal restatement of the reference."""
    rng = np.random.default_rng(2024)
    for it in range(40):
        V = int(rng.integers(1, 40))
        E = int(rng.integers(0, 4 * V + 1))
        if it % 7 == 0:  # a chain: deep BFS (many levels, tail handled top-down)
            V = 64
            s, d = np.arange(V - 1, dtype=np.int64), np.arange(1, V, dtype=np.int64)
        else:
            s, d = rng.integers(0, V, E), rng.integers(0, V, E)
        e = np.arange(len(s), dtype=np.int64)
        w = rng.integers(0, 9, len(s))
        st, ora = both(V, (s, d, e), w=w)
        n = int(rng.integers(1, 200))
        ps, pd = rng.integers(0, V, n), rng.integers(0, V, n)
        valid = rng.random(n) > 0.1
        pgq.set_option("words", int(rng.choice([0, 1, 2, 16])))
        pgq.set_option("streams", int(rng.integers(1, 4)))
        ln, ok = st.iterativelength(0, V, ps, pd, src_valid=valid)
        oln, ook = ora.iterativelength(V, ps, pd, src_valid=valid)
        assert lens(ln, ok) == lens(oln, ook), (it, V, E)
        assert st.shortestpath(0, V, ps, pd, src_valid=valid) == ora.shortestpath(V, ps, pd, src_valid=valid), (it, V, E)
        if len(s):
            out, cok = st.cheapest_path_length(0, V, ps[valid], pd[valid])
            lout, lok = ora.lean_cheapest_path_length(V, ps[valid], pd[valid])
            assert (cok == lok).all() and (out[cok] == lout[lok]).all(), (it, V, E)
    # zero rows, all-NULL rows
    st, ora = both(5, directed_rows(load_golden("student_directed.json")["edges"]))
    ln, ok = st.iterativelength(0, 5, np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64))
    assert len(ln) == 0
    ln, ok = st.iterativelength(0, 5, [1, 2, 3], [0, 0, 0], src_valid=[False, False, False])
    assert not ok.any() and st.shortestpath(0, 5, [1, 2], [0, 0], src_valid=[False, False]) == [None, None]


def test_concurrent_callers_share_one_csr():
    # DuckDB calls the UDFs from many worker threads over one shared read-only CSR (SURVEY §8b "Threading")
    import threading
    rng = np.random.default_rng(77)
    V, E = 20000, 150000
    s, d, e = random_graph(rng, V, E)
    w = rng.integers(1, 100, E)
    st, ora = both(V, (s, d, e), w=w)
    jobs = []
    for t in range(6):
        ps, pd = rng.integers(0, V, 1500), rng.integers(0, V, 1500)
        oln, ook = ora.lean_iterativelength(V, ps, pd, nthreads=4)
        jobs.append((ps, pd, oln, ook, ora.lean_shortestpath(V, ps[:300], pd[:300]),
                     ora.lean_cheapest_path_length(V, ps[:200], pd[:200])))
    errors = []

    def work(job):
        try:
            ps, pd, oln, ook, opaths, (cout, cok) = job
            for _ in range(3):
                ln, ok = st.iterativelength(0, V, ps, pd)
                assert (ok == ook).all() and (ln[ok] == oln[ok]).all()
                assert st.shortestpath(0, V, ps[:300], pd[:300]) == opaths
                out, okc = st.cheapest_path_length(0, V, ps[:200], pd[:200])
                assert (okc == cok).all() and (out[okc] == cout[cok]).all()
        except Exception as ex:  # noqa: BLE001
            errors.append(repr(ex))

    threads = [threading.Thread(target=work, args=(j,)) for j in jobs]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert errors == []


def test_two_ranks_on_one_gpu_product_path():
    """N > 1 with the PRODUCT on every rank: two processes share cuda:0 (gloo carries the collectives, RCCL needs one
    GPU per rank), CSR broadcast, pairs sharded, results gathered; bench.py asserts every rank's lengths against the
    accounting pass and prints one JSON line."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for extra in (["--workload", "snb_sf100", "--scaling", "strong"], ["--workload", "snb_paths", "--scaling", "weak"],
                  ["--workload", "snb_sf100"]):  # the default for N > 1 is strong (configs[3]: 65,536 pairs in total)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
               "127.0.0.1", "--master-port", "29631", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2",
               "--warmup", "1", "--backend", "gloo", "--pairs-per-gpu", "3000", "--snb-vertices", "20000",
               "--snb-friendships", "400000", "--no-cpu-baseline"] + extra
        if len(extra) == 2 and "strong" not in extra and "weak" not in extra:
            # the bare form `python bench.py --gpus 2 ...`: bench.py starts the two ranks itself (round 4 ran one)
            cmd = [sys.executable] + cmd[cmd.index(os.path.join(root, "bench.py")):]
        env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
        out = json.loads(line)
        assert out["n_gpus"] == 2 and out["value"] > 0
        assert out["config"]["pairs_total"] == (6000 if "weak" in extra else 3000)
        assert out["scaling"] == ("weak" if "weak" in extra else "strong")
        if "weak" not in extra:  # the other mode is measured in the same run
            assert out["weak"]["pairs_total"] == 6000 and out["weak"]["pairs_per_s"] > 0


def test_in_library_multi_gpu_shards_and_gathers():
    """pgq_init_devices + pgq_csr_replicate + pgq_iterativelength_multi: one host thread and one CSR replica per enabled
    device, contiguous shards, results gathered into one host array.  On a one-GPU box the device list names device 0
    twice: two replicas (peer copy onto the same device), two shard threads, two workspaces."""
    rng = np.random.default_rng(31)
    V, E = 40000, 600000
    s, d, e = random_graph(rng, V, E)
    off, adj, eid = graphgen.csr_from_rows(V, s, d)
    off_ = off
    dev = pgq.DeviceCSR(V, off, adj, eid)
    ora = OracleCSR.adopt(V, off, adj, eid)
    n = 9001
    ps, pd = rng.integers(0, V, n), rng.integers(0, V, n)
    ps[:5] = -1  # NULL rows
    oln, ook = ora.lean_iterativelength(V, np.maximum(ps, 0), pd, nthreads=8)
    want = np.where(ook & (ps >= 0), oln, -1)
    assert pgq.init_devices([0, 0]) == 2
    try:
        for meet in (1, 0):
            pgq.set_option("meet", meet)
            got = dev.iterativelength_multi(ps, pd)
            assert (got == want).all()
        # paths: every shard's ragged lists gathered behind each other, offsets shifted by the preceding shards' sizes
        m = 3001
        opaths = ora.lean_shortestpath(V, np.maximum(ps[:m], 0), pd[:m])
        for meet in (1, 0):
            pgq.set_option("meet", meet)
            ln, off, child = dev.shortestpath_multi(ps[:m], pd[:m])
            got = [None if ln[i] < 0 else child[off[i]:off[i] + 2 * ln[i] + 1].tolist() for i in range(m)]
            assert got == [p if ps[i] >= 0 else None for i, p in enumerate(opaths)]
            assert len(child) == sum(len(p) for p in got if p is not None)
        # weighted: both weight types
        for wt in (np.int64, np.float64):
            w = rng.integers(1, 50, len(adj)).astype(wt)
            if wt is np.float64:
                w = w / 7.0
            devw = pgq.DeviceCSR(V, off_, adj, eid, w)
            oraw = OracleCSR.adopt(V, off_, adj, eid, w)
            k = 600
            want_w, want_ok = oraw.lean_cheapest_path_length(V, ps[5:5 + k], pd[5:5 + k])
            got_w, got_ok = devw.cheapest_path_length_multi(ps[5:5 + k], pd[5:5 + k])
            assert (got_ok == want_ok).all() and (got_w[want_ok] == want_w[want_ok]).all()
            devw.close()
    finally:
        pgq.init_devices([0])


def test_cheapest_chain_prepass_cycles_and_branches():
    """k_chain_walk: rows whose source starts an out-degree-1 chain are answered by walking it; cycles (step cap) and
    branching vertices fall back to the batched relaxation.  Same values with the pre-pass off."""
    rng = np.random.default_rng(91)
    V = 64
    # ring 0->1->...->9->0, tail 20->21->22->3 (joins the ring), chain 30->31->32 (dead end), branch at 40: 40->41, 40->42->43
    edges = [(i, (i + 1) % 10) for i in range(10)] + [(20, 21), (21, 22), (22, 3), (30, 31), (31, 32), (40, 41), (40, 42), (42, 43)]
    s = np.array([e[0] for e in edges]); d = np.array([e[1] for e in edges])
    off, adj, eid = graphgen.csr_from_rows(V, s, d)
    for w in (rng.integers(1, 50, len(adj)), rng.random(len(adj)) + 0.5):
        dev = pgq.DeviceCSR(V, off, adj, eid, w)
        ora = OracleCSR.adopt(V, off, adj, eid, w)
        ps, pd = np.repeat(np.arange(V), V), np.tile(np.arange(V), V)
        want, wok = ora.lean_cheapest_path_length(V, ps, pd)
        for chain, cap in ((1, 4096), (1, 3), (0, 4096)):
            pgq.set_option("chain", chain)
            pgq.set_option("chain_cap", cap)
            out, ok = dev.cheapest_path_length(ps, pd)
            assert (ok == wok).all() and (out[ok] == want[wok]).all()


def test_local_clustering_coefficient_device_bit_exact():
    """k_lcc / k_lcc_big against the reference's goldens (student graph, SNB SF0.003) and the literal restatement on a
    skewed graph with adjacency lists beyond the 512-entry hash table (bit map path); float32 results bit for bit."""
    from test_oracle_golden import repr_float32
    lcc = load_golden("lcc.json")
    g = lcc["student"]
    st, _ = both(g["V"], undirected_rows(g["edges"]))
    out, ok = st.local_clustering_coefficient(0, np.arange(g["V"]))
    assert ok.all() and [repr_float32(x) for x in out] == [r[1] for r in g["rows"]]
    snb = load_golden("snb003_knows.json")
    st, _ = both(snb["V"], undirected_rows(snb["edges"]), csr_id=1)
    ids = np.array([r[0] for r in lcc["snb003"]["rows"]])
    out, ok = st.local_clustering_coefficient(1, ids)
    assert [repr_float32(x) for x in out] == [r[1] for r in lcc["snb003"]["rows"]]
    rng = np.random.default_rng(44)
    V, E = 8000, 300000
    s, d, e = random_graph(rng, V, E, skew=True)  # parallel edges, self loops, lists of several thousand entries
    off, adj, eid = graphgen.csr_from_rows(V, s, d)
    assert np.diff(off).max() > 600
    dev = pgq.DeviceCSR(V, off, adj, eid)
    ora = OracleCSR.adopt(V, off, adj, eid)
    ids = np.arange(V)
    valid = rng.random(V) > 0.02
    out, ok = dev.local_clustering_coefficient(ids, src_valid=valid)
    want = ora.local_clustering_coefficient(ids)
    assert (ok == valid).all() and (out[valid].view(np.uint32) == want[valid].view(np.uint32)).all()
    with pytest.raises(pgq.PgqError, match="out of range"):
        dev.local_clustering_coefficient(np.array([V + 3]))


def test_pagerank_device_matches_reference():
    """pgq_pagerank against the reference's goldens and the literal restatement (V + 2 entries, dangling vertices, the
    same iteration count); tolerance 1e-12 relative: only the dangling-rank total is summed in a different order."""
    pr = load_golden("pagerank.json")
    for k, key in enumerate(("g1", "g2")):
        g = pr[key]
        st, ora = both(g["V"], directed_rows(g["edges"]), csr_id=k)
        out, ok = st.pagerank(k, np.arange(g["V"]))
        assert ok.all()
        for vid, txt in g["rows"]:
            assert abs(out[vid] - float(txt)) <= 1e-12 * float(txt), (key, vid, out[vid], txt)
    rng = np.random.default_rng(45)
    V, E = 20000, 150000
    s, d, e = random_graph(rng, V, E, skew=True)  # many dangling vertices
    off, adj, eid = graphgen.csr_from_rows(V, s, d)
    dev = pgq.DeviceCSR(V, off, adj, eid)
    want, wit = OracleCSR.adopt(V, off, adj, eid).pagerank()
    ids = np.concatenate([np.arange(V + 2), [-1, V + 2]])
    out, ok, it = dev.pagerank(ids)
    assert ok[:V + 2].all() and not ok[V + 2:].any() and it == wit
    assert np.max(np.abs(out[:V + 2] - want) / want) <= 1e-12


@pytest.mark.parametrize("cfg", ["lane_batches", "prepass"])
def test_unpinned_variants_1500_rows_against_the_oracle(cfg):
    """iterativelength2, iterativelengthbidirectional and reachability have no test in the reference (parity unpinned):
    1500 random rows with NULL sources on a skewed graph and on a sparse one (dead ends, unreachable pairs) against the
    oracle's restatement of iterativelength2.cpp:33-130 and the hop counts of iterativelength.cpp:34-143 — through the
    lane batches and through the pre-pass."""
    for k, v in PRODUCT_CONFIGS[cfg].items():
        pgq.set_option(k, v)
    rng = np.random.default_rng(808)
    for V, E, skew in ((3000, 30000, True), (4000, 5000, False)):
        st, ora = both(V, random_graph(rng, V, E, skew=skew))
        n = 1500
        ps, pd = rng.integers(0, V, n), rng.integers(0, V, n)
        ps[:40] = pd[:40]
        valid = rng.random(n) > 0.05
        oln, ook = ora.iterativelength(V, ps, pd, src_valid=valid)
        o2, o2k = ora.iterativelength(V, ps, pd, src_valid=valid, variant=2)
        assert lens(o2, o2k) == lens(oln, ook)  # the two restatements agree with each other
        for variant in (1, 2, 3):
            ln, ok = st.iterativelength(0, V, ps, pd, src_valid=valid, variant=variant)
            assert lens(ln, ok) == lens(oln, ook), variant
        r, rok = st.reachability(0, V, ps, pd, src_valid=valid)
        assert (rok == valid).all()  # NULL source -> NULL (intentional deviation from the reference)
        assert (r[valid] == ook[valid]).all()


@pytest.mark.slow
def test_c5_forest_at_bench_scale_2_24():
    """BASELINE configs[4] at the scale bench.py runs it (reply forest, V = 2^24): 4096 pairs with ancestor destinations
    (and some unreachable ones), int64 and double weights, every value against the oracle's Dijkstra."""
    rng = np.random.default_rng(56)
    V, s, d = graphgen.reply_forest(1 << 24, seed=5)
    off, adj, eid = graphgen.csr_from_rows(V, s, d)
    deg = np.diff(off)
    cand = np.nonzero(deg > 0)[0]
    src = cand[rng.integers(0, len(cand), 4096)]
    dst = src.copy()
    hops = rng.integers(1, 9, 4096)
    for h in range(8):
        move = (hops > h) & (deg[dst] > 0)
        dst[move] = adj[off[dst[move]]]
    dst[:512] = rng.integers(0, V, 512)
    for w in (rng.integers(1, 1000, len(adj)), rng.random(len(adj)) + 0.01):
        dev = pgq.DeviceCSR(V, off, adj, eid, w)
        ora = OracleCSR.adopt(V, off, adj, eid, w)
        out, ok = dev.cheapest_path_length(src, dst)
        want, wok = ora.lean_cheapest_path_length(V, src, dst)
        assert (ok == wok).all() and (out[ok] == want[wok]).all() and ok.sum() > 3000
        dev.close()


def test_eight_shards_on_one_gpu_65536_pairs():
    """configs[3] as the in-library multi-GPU call sees it: 65,536 pairs cut into 8 shards (device 0 named eight times:
    eight replicas, eight shard threads), every length against the single-shard answer and a sample against the oracle."""
    import time
    rng = np.random.default_rng(65)
    V2, s2, d2 = graphgen.snb_knows_like(60000, 1500000, seed=9)
    off, adj, eid = graphgen.csr_from_rows(V2, s2, d2)
    dev = pgq.DeviceCSR(V2, off, adj, eid)
    ora = OracleCSR.adopt(V2, off, adj, eid)
    n = 65536
    ps, pd = rng.integers(0, V2, n), rng.integers(0, V2, n)
    pgq.set_option("meet", 1)
    one = dev.iterativelength_multi(ps, pd)  # device list [0]: one shard
    oln, ook = ora.lean_iterativelength(V2, ps[:2000], pd[:2000], nthreads=8)
    assert (one[:2000] == np.where(ook, oln, -1)).all()
    assert pgq.init_devices([0] * 8) == 8
    try:
        got = dev.iterativelength_multi(ps, pd)
        assert (got == one).all()
        t0 = time.perf_counter()
        for _ in range(5):
            got = dev.iterativelength_multi(ps, pd)
        dt8 = (time.perf_counter() - t0) / 5
        assert (got == one).all()
    finally:
        pgq.init_devices([0])
    t0 = time.perf_counter()
    for _ in range(5):
        dev.iterativelength_multi(ps, pd)
    dt1 = (time.perf_counter() - t0) / 5
    # eight shards on ONE device cannot be faster than one; what is checked is that sharding adds little fixed cost
    print("65536 pairs: 1 shard %.3f ms, 8 shards on one GPU %.3f ms" % (dt1 * 1e3, dt8 * 1e3))
    assert dt8 < dt1 + 0.004  # < 0.5 ms of fixed cost per shard, host copies included


def test_replicas_follow_the_enabled_device_list():
    """a *_multi call before pgq_init_devices leaves a one-entry replica list; the next call with more devices
    enabled must rebuild it instead of indexing past its end."""
    rng = np.random.default_rng(66)
    V, E = 5000, 40000
    s, d, e = random_graph(rng, V, E)
    off, adj, eid = graphgen.csr_from_rows(V, s, d)
    dev = pgq.DeviceCSR(V, off, adj, eid)
    ps, pd = rng.integers(0, V, 3000), rng.integers(0, V, 3000)
    a = dev.iterativelength_multi(ps, pd)  # replicas built for [0]
    assert pgq.init_devices([0, 0, 0]) == 3
    try:
        b = dev.iterativelength_multi(ps, pd)  # three shards: replicas must be extended
        assert (a == b).all()
    finally:
        pgq.init_devices([0])
    c = dev.iterativelength_multi(ps, pd)
    assert (a == c).all()


def test_weakly_connected_component_device_matches_goldens_and_oracle():
    """weakly_connected_component on the device (spanning forest under the reference's processing order by Boruvka rounds,
    then the reference's own Link over its edges): the component id is the root the reference's sequential union-find
    ends in (goldens: id 2 for the cycle 0-1-2-3), not a canonical label."""
    for case in load_golden("wcc.json")["cases"]:  # weakly_connected_component.test
        s, d, e = undirected_rows(case["edges"])
        st = pgq.PgqState()
        st.build_csr(0, case["V"], s, d, e)
        out, ok = st.weakly_connected_component(0, np.arange(case["V"]))
        assert ok.all() and [[i, int(c)] for i, c in enumerate(out)] == case["rows"], case["source"]
    rng = np.random.default_rng(6)
    graphs = []
    V, E = 3000, 2500  # sparse: many components
    s, d = rng.integers(0, V, E), rng.integers(0, V, E)
    graphs.append((V,) + undirected_rows(np.stack([s, d], axis=1)))
    V = 5000  # a directed chain and a reversed one: every vertex hooks onto its neighbour (deep hook chains)
    graphs.append((V, np.arange(V - 1, dtype=np.int64), np.arange(1, V, dtype=np.int64), np.arange(V - 1, dtype=np.int64)))
    graphs.append((V, np.arange(1, V, dtype=np.int64), np.arange(V - 1, dtype=np.int64), np.arange(V - 1, dtype=np.int64)))
    graphs.append((4000,) + random_graph(rng, 4000, 30000, skew=True))  # directed, duplicates, self loops, hubs
    graphs.append((3, np.array([1], dtype=np.int64), np.array([1], dtype=np.int64), np.zeros(1, dtype=np.int64)))  # one self loop
    V2, s2, d2 = graphgen.reply_forest(1 << 17, seed=3)
    graphs.append((V2, s2, d2, np.arange(len(s2), dtype=np.int64)))
    for k, (V, us, ud, ue) in enumerate(graphs):
        st = pgq.PgqState()
        st.build_csr(k, V, us, ud, ue)
        # the two trailing forest entries are accepted like the reference does (V: its own root; V + 1: the root of vertex
        # 0, the zero a resize left there); NULL and out-of-range ids give NULL
        ids = np.concatenate([np.arange(V + 2), [-1, V + 2, V + 50]])
        out, ok = st.weakly_connected_component(k, ids)
        want, wok = OracleCSR.from_edges(V, us, ud, ue).weakly_connected_component(ids)
        assert (ok == wok).all() and ok[:V + 2].all() and not ok[V + 2:].any(), k
        assert (out[ok] == want[wok]).all(), k
    with pytest.raises(pgq.PgqError, match="CSR not found. Is the graph populated"):
        st.weakly_connected_component(99, [0])


def test_options_of_one_handle_do_not_leak_into_another():
    """pgq_csr_set_option: a handle's own copy of the options (round 2's were process-wide only: two connections, or
    a test, tuned each other's searches)."""
    import torch
    rng = np.random.default_rng(17)
    V, E = 20000, 200000
    s, d, e = random_graph(rng, V, E)
    off, adj, eid = graphgen.csr_from_rows(V, s, d)
    a, b = pgq.DeviceCSR(V, off, adj, eid), pgq.DeviceCSR(V, off, adj, eid)
    ora = OracleCSR.adopt(V, off, adj, eid)
    pgq.set_option("meet", 1)
    a.set_option("meet", 0)           # handle a: lane batches only, with another batch width
    a.set_option("words", 2)
    assert a.get_option("meet") == 0 and b.get_option("meet") == 1 and pgq.get_option("meet") == 1
    n = 3000
    ps, pd = rng.integers(0, V, n), rng.integers(0, V, n)
    oln, ook = ora.lean_iterativelength(V, ps, pd, nthreads=4)
    want = np.where(ook, oln, -1)
    d_src, d_dst = torch.from_numpy(ps).cuda(), torch.from_numpy(pd).cuda()
    d_len = torch.empty(n, dtype=torch.int64, device="cuda")
    for dev, expect_meet in ((a, False), (b, True), (a, False)):
        pgq.reset_stats()
        dev.iterativelength_bulk_ptr(n, d_src.data_ptr(), d_dst.data_ptr(), d_len.data_ptr())
        assert (d_len.cpu().numpy() == want).all()
        st = pgq.get_stats()
        assert (st["meet_pairs"] > 0) == expect_meet  # handle a never sees the pre-pass, handle b does
        if not expect_meet:
            assert st["levels"] > 0
    with pytest.raises(pgq.PgqError, match="unknown option"):
        a.set_option("no_such_option", 1)


# ---- round 5: levels enqueued ahead of the host, rows left in place, the route memo --------------------------------------

def _ring_with_chords(V, chords, rng):
    a = np.arange(V, dtype=np.int64)
    s = np.concatenate([a, (a + 1) % V, rng.integers(0, V, chords)])
    d = np.concatenate([(a + 1) % V, a, rng.integers(0, V, chords)])
    return s, d, np.arange(len(s), dtype=np.int64)


def test_levels_enqueued_ahead_match_the_round_trip_loop():
    """spec_levels: the second batch of a width runs its levels under the first one's plan, k_level_reset checking each on
    the device.  Same answers as one host round trip per level (spec_levels = 0) and as the oracle: a plan that fits, a plan
    the level rule contradicts (called off at level 1: force_mode flips between the calls), a plan that runs out (a ring:
    more levels than the log holds, and a second call that needs MORE levels than the first)."""
    rng = np.random.default_rng(55)
    V, E = 20000, 160000
    st, ora = both(V, random_graph(rng, V, E, skew=True))
    n = 5000
    ps, pd = rng.integers(0, V, n), rng.integers(0, V, n)
    oln, ook = ora.lean_iterativelength(V, ps, pd, nthreads=8)
    want = lens(oln, ook)
    pgq.set_option("spec_levels", 0)
    pgq.reset_stats()
    ln, ok = st.iterativelength(0, V, ps, pd)
    assert lens(ln, ok) == want
    waits_round_trip = pgq.get_stats()["host_waits"]
    pgq.set_option("spec_levels", 1)
    for rep in range(3):
        pgq.reset_stats()
        ln, ok = st.iterativelength(0, V, ps, pd)
        assert lens(ln, ok) == want
        stats = pgq.get_stats()
        assert stats["spec_batches"] >= 1  # (a batch much narrower than the one that left the plan may be called off at level 1)
    assert stats["spec_levels"] >= 1 and stats["host_waits"] < waits_round_trip
    # the level rule changes under the plan: every level is top-down now, the plan's bottom-up levels are called off
    pgq.set_option("force_mode", 1)
    pgq.reset_stats()
    ln, ok = st.iterativelength(0, V, ps, pd)
    assert lens(ln, ok) == want and pgq.get_stats()["spec_aborts"] >= 1
    pgq.set_option("force_mode", 2)
    pgq.reset_stats()
    ln, ok = st.iterativelength(0, V, ps, pd)
    assert lens(ln, ok) == want and pgq.get_stats()["spec_aborts"] >= 1
    pgq.set_option("force_mode", 0)
    # a long-diameter graph: 150 levels > the 61 the log holds; then pairs that need more levels than the plan has
    V2 = 300
    st2, ora2 = both(V2, _ring_with_chords(V2, 0, rng))
    near = np.arange(40, dtype=np.int64)
    for src, dst in ((near, (near + 5) % V2), (near, (near + 140) % V2), (near, (near + 5) % V2), (near, (near + 149) % V2)):
        ln, ok = st2.iterativelength(0, V2, src, dst)
        o1, o2 = ora2.lean_iterativelength(V2, src, dst)
        assert lens(ln, ok) == lens(o1, o2)
    # traversed-edge accounting and paths stay on the round-trip loop and still agree
    assert st.shortestpath(0, V, ps[:800], pd[:800]) == ora.lean_shortestpath(V, ps[:800], pd[:800])


def test_route_memo_follows_the_rows():
    """Large calls (> 16,384 rows) on the same buffers: a cross product is sent to the lane batches by the sampled
    decision once and goes there straight afterwards (no pre-pass chain); when the SAME device buffers then hold
    scattered pairs the sample taken beside the lane assignment says so and the call after it runs the pre-pass again.
    Every answer equal to the oracle's whatever the route."""
    import torch
    rng = np.random.default_rng(66)
    V, E = 30000, 600000
    s, d, e = random_graph(rng, V, E)
    off, adj, eid = graphgen.csr_from_rows(V, s, d)
    dev = pgq.DeviceCSR(V, off, adj, eid)
    ora = OracleCSR.adopt(V, off, adj, eid)
    pgq.set_option("meet", 1)
    n = 40000
    srcs = rng.choice(V, 40, replace=False)
    cross = np.stack([np.repeat(srcs, n // 40), rng.integers(0, V, n)], axis=1).astype(np.int64)
    scattered = rng.integers(0, V, (n, 2)).astype(np.int64)
    buf = torch.empty((n, 2), dtype=torch.int64, device="cuda")
    out = torch.empty(n, dtype=torch.int64, device="cuda")
    d_src = torch.empty(n, dtype=torch.int64, device="cuda")
    d_dst = torch.empty(n, dtype=torch.int64, device="cuda")
    routes = []
    for rows in (cross, cross, cross, scattered, scattered, scattered, cross, cross):
        t = torch.from_numpy(rows).to("cuda")
        d_src.copy_(t[:, 0])
        d_dst.copy_(t[:, 1])
        torch.cuda.synchronize()
        pgq.reset_stats()
        dev.iterativelength_bulk_ptr(n, d_src.data_ptr(), d_dst.data_ptr(), out.data_ptr())
        got = out.cpu().numpy()
        oln, ook = ora.lean_iterativelength(V, rows[:, 0], rows[:, 1], nthreads=8)
        assert ((got >= 0) == ook).all() and (got[ook] == oln[ook]).all()
        routes.append("prepass" if pgq.get_stats()["meet_pairs"] > 0 else "lanes")
    assert routes[:3] == ["lanes"] * 3  # the sample calls the pre-pass off; then the memo skips its chain
    assert routes[4:6] == ["prepass"] * 2  # one call late at most: the memo'd route is a matter of speed, not of answers
    assert routes[7] == "lanes"
    del buf


def test_zero_copy_chunk_path_with_selection_validity_and_null_rows():
    """The chunk entry point resolves DuckDB's vectors straight into the pinned staging block the pre-pass kernels read
    (chunk_zero_copy = 1): selection vectors on both sides, NULL sources, rows whose (ignored) destination payload is out
    of range behind a NULL source — against the copy path (chunk_zero_copy = 0) and the oracle."""
    rng = np.random.default_rng(77)
    V, E = 8000, 90000
    st, ora = both(V, random_graph(rng, V, E, skew=True))
    pgq.set_option("meet", 1)
    pgq.set_option("meet_bias", 1e9)
    base_s, base_d = rng.integers(0, V, 700), rng.integers(0, V, 900)
    n = 2048
    ssel, dsel = rng.integers(0, 700, n).astype(np.uint32), rng.integers(0, 900, n).astype(np.uint32)
    valid = rng.random(700) > 0.1
    base_s = base_s.copy()
    base_s[~valid] = 2 ** 40  # garbage payload under a NULL: must not be looked at
    ps, pd = np.where(valid[ssel], base_s[ssel], 0), base_d[dsel]
    oln, ook = ora.lean_iterativelength(V, ps, pd)
    want = [int(v) if (k and vv) else None for v, k, vv in zip(oln, ook, valid[ssel])]
    res = {}
    for zc in (1, 0):
        pgq.set_option("chunk_zero_copy", zc)
        ln, ok = st.iterativelength(0, V, base_s, base_d, src_valid=valid, src_sel=ssel, dst_sel=dsel)
        assert lens(ln, ok) == want
        res[zc] = (ln.copy(), ok.copy())
    assert (res[0][0] == res[1][0]).all() and (res[0][1] == res[1][1]).all()
    # a constant vector (selection of zeros) on the source side: the binder's cross product arrives like this
    zsel = np.zeros(n, dtype=np.uint32)
    ln, ok = st.iterativelength(0, V, base_s[valid][:1], base_d, src_sel=zsel, dst_sel=dsel)
    o1, o2 = ora.lean_iterativelength(V, np.full(n, base_s[valid][0]), base_d[dsel])
    assert lens(ln, ok) == lens(o1, o2)


def test_weighted_ring_with_chords_takes_plain_rounds_and_stays_fast():
    """Round 4 shipped relax_light = 1 for every CSR: a weighted ring with chords (20,000 vertices, three or four edges
    each: nothing for the weight cap to skip) took 1830 rounds / 0.28 s per 64 pairs against 365 rounds / 0.05 s with plain
    rounds.  The shipped rule now goes by the mean out-degree: same values as Dijkstra, int64 and double, within a time
    bound, and the forced light path (relax_light = 2) still agrees."""
    import time
    rng = np.random.default_rng(88)
    V = 20000
    s, d, e = _ring_with_chords(V, 4000, rng)
    for dtype in ("int64", "double"):
        w = rng.integers(1, 1000, len(s))
        if dtype == "double":
            w = w.astype(np.float64) / 7.0
        st, ora = both(V, (s, d, e), w)
        ps, pd = rng.integers(0, V, 64), rng.integers(0, V, 64)
        want, wok = ora.lean_cheapest_path_length(V, ps, pd)
        for light in (pgq.get_default_option("relax_light"), 2):
            pgq.set_option("relax_light", int(light))
            st.cheapest_path_length(0, V, ps, pd)  # warm: weight-sorted lists, label arrays
            t0 = time.perf_counter()
            out, ok = st.cheapest_path_length(0, V, ps, pd)
            dt = time.perf_counter() - t0
            assert (ok == wok).all() and (out[ok] == want[wok]).all()
            if light != 2:
                assert dt < 0.15, "the shipped rule must not take the light-edges-first path here (%.3f s)" % dt


def test_workspace_reuse_across_widths_graphs_and_entry_points_fuzz():
    """Round 5 keeps state between calls that round 4 rebuilt every time: the sparse frontier pool is cleaned by its nz
    (and zeroed whole only when its layout changes), level plans and the route memo live on the CSR handle, the
    open-lane copies are folded by the next user.  One thread (one pooled workspace) alternates graphs of different V,
    batch widths 1..32, one-batch and multi-batch calls, cross products and scattered pairs, iterativelength /
    shortestpath / the accounting pass / the bidirectional entry point — every answer against the oracle."""
    rng = np.random.default_rng(2025)
    graphs = []
    for V, E, skew in ((1, 0, False), (77, 300, False), (1000, 9000, True), (4097, 30000, False), (20000, 150000, True)):
        rows = random_graph(rng, V, E, skew=skew) if E else (np.zeros(0, np.int64), np.zeros(0, np.int64), np.zeros(0, np.int64))
        off, adj, eid = graphgen.csr_from_rows(V, rows[0], rows[1])
        graphs.append((V, pgq.DeviceCSR(V, off, adj, eid), OracleCSR.adopt(V, off, adj, eid)))
    pgq.set_option("meet", 0)  # every row through the lane batches (the pre-pass has its own tests)
    for it in range(60):
        V, dev, ora = graphs[int(rng.integers(0, len(graphs)))]
        shape = int(rng.integers(0, 4))
        if shape == 0:  # scattered pairs: many sources
            n = int(rng.integers(1, 6000))
            ps, pd = rng.integers(0, V, n), rng.integers(0, V, n)
        elif shape == 1:  # cross product: few sources, rows grouped
            k = int(rng.integers(1, 40))
            per = int(rng.integers(1, 300))
            ps = np.repeat(rng.integers(0, V, k), per)
            pd = rng.integers(0, V, len(ps))
        elif shape == 2:  # one source x every vertex
            ps, pd = np.full(V, int(rng.integers(0, V))), np.arange(V, dtype=np.int64)
        else:  # a handful of rows
            n = int(rng.integers(1, 70))
            ps, pd = rng.integers(0, V, n), rng.integers(0, V, n)
        pgq.set_option("words", int(rng.choice([0, 0, 1, 2, 8, 32])))
        pgq.set_option("spec_levels", int(rng.integers(0, 2)))
        pgq.set_option("force_mode", int(rng.choice([0, 0, 0, 1, 2])))
        pgq.set_option("streams", int(rng.integers(1, 4)))
        oln, ook = ora.lean_iterativelength(V, ps, pd)
        what = int(rng.integers(0, 5))
        if what <= 2:
            for rep_ in range(1 + int(rng.integers(0, 2))):  # now and then the same call again: stage 2 ahead of the wait, the level plan
                ln, ok = dev.iterativelength(ps, pd)
                assert lens(ln, ok) == lens(oln, ook), (it, V, shape)
        elif what == 3:
            m = min(len(ps), 400)
            assert dev.shortestpath(ps[:m], pd[:m]) == ora.lean_shortestpath(V, ps[:m], pd[:m]), (it, V, shape)
        else:
            ln, ok = dev.iterativelength_bidirectional(ps, pd) if hasattr(dev, "iterativelength_bidirectional") else dev.iterativelength(ps, pd)
            assert lens(ln, ok) == lens(oln, ook), (it, V, shape)
    for _, dev, _ in graphs:
        dev.close()


def test_cheapest_path_label_width_switches_at_31_bits():
    """int64 weights keep 4-byte labels while w_max x V stays under 2^31 - 1 and 8-byte labels above (round 5): long chains of
    maximal weights on both sides of the boundary, so that real labels come close to what the narrow form can hold; both
    against the oracle's Dijkstra, and the narrow form against the wide one (relax_labels32 = 0)."""
    rng = np.random.default_rng(91)
    V = 3000
    a = np.arange(V - 1, dtype=np.int64)
    for w_max in (715000, 716500, 10 ** 12):  # 715000 x 3000 = 2.145e9 < 2^31 - 1 < 716500 x 3000
        s = np.concatenate([a, rng.integers(0, V, 40000)])
        d = np.concatenate([a + 1, rng.integers(0, V, 40000)])
        w = np.concatenate([np.full(V - 1, w_max, dtype=np.int64), rng.integers(w_max // 2, w_max + 1, 40000)])
        st, ora = both(V, (s, d, np.arange(len(s), dtype=np.int64)), w=w)
        ps = np.concatenate([np.zeros(40, dtype=np.int64), rng.integers(0, V, 200)])
        pd = np.concatenate([np.full(40, V - 1, dtype=np.int64) - np.arange(40), rng.integers(0, V, 200)])
        want, wok = ora.lean_cheapest_path_length(V, ps, pd)
        got = {}
        for narrow in (1, 0):
            pgq.set_option("relax_labels32", narrow)
            out, ok = st.cheapest_path_length(0, V, ps, pd)
            assert (ok == wok).all() and (out[ok] == want[wok]).all(), (w_max, narrow)
            got[narrow] = out
        assert (got[0] == got[1]).all()


def test_lazy_edge_ids_cross_pcie_on_the_first_call_that_reads_them():
    # PGQ_UPLOAD_LAZY_EDGE_IDS (pgq_csr_upload_ex): iterativelength never touches edge ids, so the UDF layer leaves them on
    # the host; the first shortestpath / download / replicate brings them over.  Same lists as an eager upload.
    rng = np.random.default_rng(91)
    V, E = 5000, 60000
    s, d, _ = random_graph(rng, V, E)
    eid = rng.permutation(E).astype(np.int64) + 1000  # not the slot index: a wrong or missing copy shows
    off, adj, e2 = graphgen.csr_from_rows(V, s, d)
    ids = eid[e2]
    eager = pgq.DeviceCSR(V, off, adj, ids)
    lazy = pgq.DeviceCSR(V, off, adj, ids, lazy_edge_ids=True)
    ps, pd = rng.integers(0, V, 600), rng.integers(0, V, 600)
    a, ok_a = eager.iterativelength(ps, pd)
    b, ok_b = lazy.iterativelength(ps, pd)
    assert (ok_a == ok_b).all() and (a == b).all()
    bytes_before = lazy.device_bytes
    want = eager.shortestpath(ps, pd)
    assert lazy.shortestpath(ps, pd) == want and any(p is not None and len(p) > 1 for p in want)
    assert lazy.device_bytes == bytes_before + 8 * len(adj)  # the ids were copied by that call
    assert lazy.shortestpath(ps, pd) == want
    _, _, got_ids, _ = lazy.download()
    assert (got_ids == ids).all()
    lazy2 = pgq.DeviceCSR(V, off, adj, ids, lazy_edge_ids=True)  # download first, then paths
    assert (lazy2.download()[2] == ids).all() and lazy2.shortestpath(ps[:50], pd[:50]) == want[:50]
    for c in (eager, lazy, lazy2):
        c.close()


@pytest.mark.parametrize("cache", [1, 0])
def test_first_call_on_a_fresh_handle_equals_the_tenth(cache):
    # The reference's CSR lives for ONE query (iterative_length_function_data.cpp:27, duckpgq_state.cpp:162-170): a handle's
    # first call — no route memo, no level plan, no measured bytes per row; with calibration_cache = 1 whatever an earlier
    # handle over the same graph shape left behind — must give what its tenth call gives, on every route.
    import torch
    rng = np.random.default_rng(17)
    V, E = 30000, 600000
    s, d, _ = random_graph(rng, V, E)
    off, adj, eid = graphgen.csr_from_rows(V, s, d)
    ora = OracleCSR.adopt(V, off, adj, eid)
    for k in ("meet", "ball", "wbibfs"):
        pgq.set_option(k, pgq.get_default_option(k))
    pgq.set_option("calibration_cache", cache)
    pgq.set_option("ball_seg_kb", 16)
    srcs = rng.choice(V, 30, replace=False)
    shapes = {"scattered": (rng.integers(0, V, 20000), rng.integers(0, V, 20000)),
              "grouped": (np.repeat(srcs, 700), rng.integers(0, V, 21000)),
              "few_sources_all_vertices": (np.repeat(srcs[:2], V), np.tile(np.arange(V, dtype=np.int64), 2)),
              "chunk": (rng.integers(0, V, 2048), rng.integers(0, V, 2048))}
    try:
        for name, (ps, pd) in shapes.items():
            oln, ook = ora.lean_iterativelength(V, ps, pd, nthreads=4)
            want = np.where(ook, oln, -1)
            t_s, t_d = torch.from_numpy(ps.astype(np.int64)).cuda(), torch.from_numpy(pd.astype(np.int64)).cuda()
            for handle in range(3):  # three fresh handles: the 2nd and 3rd find the 1st's calibration when the cache is on
                dev = pgq.DeviceCSR(V, off, adj, eid)
                t_o = torch.full((len(ps),), -7, dtype=torch.int64, device="cuda")
                for call in range(10 if handle == 0 else 2):
                    t_o.fill_(-7)
                    dev.iterativelength_bulk_ptr(len(ps), t_s.data_ptr(), t_d.data_ptr(), t_o.data_ptr())
                    assert (t_o.cpu().numpy() == want).all(), (name, handle, call)
                dev.close()
    finally:
        pgq.set_option("calibration_cache", 1)


def test_host_upload_narrows_and_range_checks_every_position():
    # pgq_csr_upload stages the pageable arrays through pinned blocks on several threads and narrows the adjacency to int32 on
    # the way (AVX2 where the host has it): the device copy must equal the host arrays, and an id outside [0, V) anywhere —
    # first element, inside a vector of eight, the scalar tail, another 4-MB block; negative, == V, >= 2^32 — must be refused.
    rng = np.random.default_rng(5150)
    V, E = 70000, 2_300_003  # more than two 4-MB blocks of int32, a tail that is not a multiple of eight
    s = np.sort(rng.integers(0, V, E))
    d = rng.integers(0, V, E)
    off = np.zeros(V + 1, dtype=np.int64)
    np.cumsum(np.bincount(s, minlength=V), out=off[1:])
    adj = d.astype(np.int64)
    eid = rng.permutation(E).astype(np.int64)
    for threads in (1, 3, 8):
        pgq.set_option("upload_threads", threads)
        dev = pgq.DeviceCSR(V, off, adj, eid)
        o2, a2, e2, _ = dev.download()
        assert (o2 == off).all() and (a2 == adj).all() and (e2 == eid).all()
        dev.close()
    pgq.set_option("upload_threads", int(pgq.get_default_option("upload_threads")))
    for pos in (0, 5, 1_048_576 + 3, E - 1, E - 9):
        for bad in (-1, V, V + 12345, 1 << 32, (1 << 40) + 7, -(1 << 35)):
            broken = adj.copy()
            broken[pos] = bad
            with pytest.raises(pgq.PgqError):
                pgq.DeviceCSR(V, off, broken, eid)
