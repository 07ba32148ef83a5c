"""CPU-only checks of the product's host side: the C-ABI libraries load and export every declared symbol, the host
mirror of create_csr_vertex/create_csr_edge reproduces the reference's CSR goldens, error texts match, and search
calls fail loudly without a GPU (no CPU fallback exists)."""
import os
import re

import numpy as np
import pytest

import duckpgq_extension_amd as pgq
from helpers import directed_rows, load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(pgq_[a-z0-9_]+)\s*\(", txt)))


@pytest.mark.parametrize("header,loader", [("pgq_hip.h", "load_hip"), ("pgq_udf.h", "load_udf")])
def test_abi_exports_every_declared_symbol(header, loader):
    lib = getattr(pgq, loader)()
    syms = declared_symbols(header)
    assert len(syms) >= 10
    missing = [s for s in syms if not hasattr(lib, s)]
    assert missing == []


@pytest.mark.parametrize("header,libname", [("pgq_hip.h", "libpgq_hip.so"), ("pgq_udf.h", "libpgq_udf.so")])
def test_abi_declares_every_exported_symbol(header, libname):
    """The other direction: every pgq_* function a library exports is declared in its public header (a definition whose
    prototype was lost in a header edit still links, loads and passes the test above)."""
    import subprocess
    so = os.path.join(ROOT, "duckpgq-extension_amd", "csrc", libname)
    out = subprocess.run(["nm", "-D", "--defined-only", so], capture_output=True, text=True, check=True).stdout
    exported = sorted({l.split()[-1] for l in out.splitlines() if " T " in l and l.split()[-1].startswith("pgq_")})
    assert len(exported) >= 10
    declared = set(declared_symbols(header))
    if header == "pgq_udf.h":  # the host mirror re-exports nothing of the device library
        exported = [s for s in exported if s.startswith("pgq_udf_") or s.startswith("pgq_state_")]
    assert [s for s in exported if s not in declared] == []


def test_host_csr_build_matches_getpgschema_golden():
    g = load_golden("student_csr_layout.json")  # getpgschema.test:85-107
    st = pgq.PgqState()
    s, d, e = directed_rows(g["edges"])
    assert st.build_csr(0, g["V"], s, d, e) == len(s)
    assert st.get_csr_v(0).tolist() == g["csr_v"]  # V+2 entries
    assert st.get_csr_e(0).tolist() == g["csr_e"]
    assert st.csr_get_w_type(0) == 0
    assert st.delete_csr(0) is True and st.delete_csr(0) is False  # delete_csr.test
    with pytest.raises(pgq.PgqError, match="CSR not found with ID 0"):  # get_csr_ptr.test:62-65
        st.get_csr_v(0)


def test_host_csr_segfault_counts_and_chunking():
    g = load_golden("csr_segfault.json")
    V = g["V"]
    ids = np.arange(V, dtype=np.int64)
    st = pgq.PgqState()
    st.build_csr(0, V, ids, ids, ids)  # 5000 rows -> three 2048-row chunks
    assert len(st.get_csr_v(0)) == g["count_v"] and len(st.get_csr_e(0)) == g["count_e"]
    assert (st.get_csr_e(0) == ids).all()


def test_host_weight_types_and_errors():
    g = load_golden("csr_w_type.json")
    lay = load_golden("student_csr_layout.json")
    s, d, e = directed_rows(lay["edges"])
    st = pgq.PgqState()
    st.build_csr(0, 5, s, d, e)
    st.build_csr(1, 5, s, d, e, w=np.full(len(s), 12, dtype=np.int64))
    st.build_csr(2, 5, s, d, e, w=np.full(len(s), 1.2))
    want = [c["value"] for c in g["cases"]]
    assert [st.csr_get_w_type(i) for i in range(3)] == want
    with pytest.raises(pgq.PgqError, match="CSR not found with ID 3"):
        st.csr_get_w_type(3)
    assert st.bind_cheapest(1) == 1 and st.bind_cheapest(2) == 2
    with pytest.raises(pgq.PgqError, match="Need to initialize CSR before doing cheapest path"):
        st.bind_cheapest(0)
    st2 = pgq.PgqState()
    st2.create_csr_vertex(0, 2, [0, 1], [1, 0])
    with pytest.raises(pgq.PgqError, match="Non-existent/non-unique vertices detected"):  # non-unique-vertices.test:40-46
        st2.create_csr_edge(0, 2, 1, 2, [0], [1], [0])
    with pytest.raises(pgq.PgqError, match="Need to initialize CSR before doing shortest path"):
        pgq.PgqState().iterativelength(0, 5, [0], [1])


def test_query_end_drops_bound_csrs():
    lay = load_golden("student_csr_layout.json")
    s, d, e = directed_rows(lay["edges"])
    st = pgq.PgqState()
    st.build_csr(0, 5, s, d, e)
    st.build_csr(7, 5, s, d, e)
    st.bind_search(0)  # IterativeLengthBind schedules deletion (iterative_length_function_data.cpp:27)
    st.query_end()
    with pytest.raises(pgq.PgqError):
        st.get_csr_v(0)
    assert len(st.get_csr_v(7)) == 7


def test_search_without_gpu_fails_loudly():
    import ctypes
    if pgq.load_hip().pgq_device_count() > 0:
        pytest.skip("a GPU is present")
    lay = load_golden("student_csr_layout.json")
    s, d, e = directed_rows(lay["edges"])
    st = pgq.PgqState()
    st.build_csr(0, 5, s, d, e)
    with pytest.raises(pgq.PgqError, match="needs a HIP device"):
        st.iterativelength(0, 5, [0], [3])
    del ctypes


def test_options_round_trip_and_unknown_key():
    """pgq_set_option / pgq_get_option share one table (no GPU needed); unknown keys are an error, not ignored."""
    import duckpgq_extension_amd as pgq
    for key, value in (("streams", 2), ("sparse_below", 2.5), ("sparse_spill", 0), ("words", 16)):
        before = pgq.get_option(key)
        pgq.set_option(key, value)
        assert pgq.get_option(key) == value
        pgq.set_option(key, int(before) if float(before).is_integer() and key != "sparse_below" else before)
        assert pgq.get_option(key) == before
    with pytest.raises(pgq.PgqError):
        pgq.set_option("no_such_knob", 1)
    with pytest.raises(pgq.PgqError):
        pgq.get_option("no_such_knob")


@pytest.mark.parametrize("seed", range(6))
def test_host_csr_build_fuzz_matches_oracle(seed):
    """create_csr_vertex + create_csr_edge of the host mirror against the oracle's restatement of
    csr_creation.cpp:14-198 on random multigraphs (self loops, parallel edges, isolated vertices, ragged chunking):
    v, e and w must be equal entry by entry — the slot order decides which parallel edge shortestpath reports."""
    from oracle.pgq_oracle import OracleCSR
    rng = np.random.default_rng(500 + seed)
    V = int(rng.integers(1, 400))
    E = int(rng.integers(0, 6000))
    s = rng.integers(0, V, E)
    d = rng.integers(0, V, E)
    if E:
        s[: E // 10] = s[0]  # one heavy source
    eid = rng.permutation(E).astype(np.int64)
    w = [None, rng.integers(0, 1000, E), rng.random(E) * 10][seed % 3]
    st = pgq.PgqState()
    st.build_csr(7, V, s, d, eid, w)
    ora = OracleCSR.from_edges(V, s, d, eid, w)
    assert st.get_csr_v(7).tolist() == ora.v.tolist()
    n = int(ora.v[V]) if V else 0
    assert st.get_csr_e(7)[:n].tolist() == ora.e[:n].tolist()
    assert st.csr_get_w_type(7) == ora.w_type
    if w is not None:
        assert st.get_csr_w(7)[:n].tolist() == ora.w[:n].tolist()


def test_host_csr_edge_chunks_from_concurrent_threads():
    """DuckDB runs create_csr_edge chunks on several worker threads (atomic slot claim, csr_creation.cpp:132-138):
    offsets must equal the single-threaded build and every vertex must hold the same multiset of (dst, edge id, w);
    only the order inside a vertex may differ.  ctypes releases the GIL, so the calls really overlap."""
    import threading
    rng = np.random.default_rng(77)
    V, E, chunk = 300, 40000, 2048
    s, d = rng.integers(0, V, E), rng.integers(0, V, E)
    s[:5000] = 3  # contended vertex
    eid = np.arange(E, dtype=np.int64)
    w = rng.integers(0, 100, E)
    ref = pgq.PgqState()
    ref.build_csr(0, V, s, d, eid, w)
    st = pgq.PgqState()
    cnt = np.bincount(s, minlength=V).astype(np.int64)
    e_sum = int(st.create_csr_vertex(0, V, np.arange(V), cnt).sum())
    chunks = [slice(lo, lo + chunk) for lo in range(0, E, chunk)]
    errors = []

    def worker(k):
        try:
            for sl in chunks[k::4]:
                st.create_csr_edge(0, V, e_sum, E, s[sl], d[sl], eid[sl], w[sl])
        except Exception as ex:  # pragma: no cover
            errors.append(ex)

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    assert errors == []
    v = st.get_csr_v(0)
    assert v.tolist() == ref.get_csr_v(0).tolist()
    e, ew, re_, rw = st.get_csr_e(0), st.get_csr_w(0), ref.get_csr_e(0), ref.get_csr_w(0)
    for x in range(V):
        lo, hi = int(v[x]), int(v[x + 1])
        assert sorted(zip(e[lo:hi].tolist(), ew[lo:hi].tolist())) == sorted(zip(re_[lo:hi].tolist(), rw[lo:hi].tolist()))


def test_udf_argument_checks_return_errors_instead_of_crashing():
    # cheapest_path_length applies the same vertex-count check as the other search UDFs (a larger V would read past
    # the host offsets); negative edge counts and a weight type that changes between chunks are rejected
    lay = load_golden("student_csr_layout.json")
    s, d, e = directed_rows(lay["edges"])
    st = pgq.PgqState()
    st.build_csr(0, 5, s, d, e, w=np.full(len(s), 3, dtype=np.int64))
    with pytest.raises(pgq.PgqError, match="vertex count does not match the CSR"):
        st.cheapest_path_length(0, 50_000_000, np.array([0]), np.array([1]))
    with pytest.raises(pgq.PgqError, match="vertex count does not match the CSR"):
        st.iterativelength(0, 50_000_000, np.array([0]), np.array([1]))
    st.create_csr_vertex(1, 5, np.arange(5), np.bincount(s, minlength=5))
    with pytest.raises(pgq.PgqError, match="negative edge count"):
        st.create_csr_edge(1, 5, -1, -1, s, d, e)
    st.create_csr_edge(1, 5, len(s), len(s), s[:3], d[:3], e[:3], w=np.ones(3, dtype=np.int64))
    with pytest.raises(pgq.PgqError, match="weight type differs"):
        st.create_csr_edge(1, 5, len(s), len(s), s[3:], d[3:], e[3:], w=np.ones(len(s) - 3, dtype=np.float64))


def test_duckdb_glue_type_checks_against_stub_headers():
    """glue/pgq_glue.cpp (replacement bodies of the search UDFs + ~CSR patch + whole-relation entry point) compiles
    against stubs of the DuckDB declarations it touches: DuckDB itself is not vendored here."""
    import subprocess
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "duckpgq-extension_amd", "csrc"), "-B", "glue-check"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_pmc_summary_reports_the_prepass_chain_per_step(tmp_path):
    """tools/pmc_summary.py: a class alias no longer averages a 900-MB k_meet3 launch with a 0.1-MB k_bibfs one (round-3
    review), and the chain's traffic is bytes x launches summed over its kernels, divided by the steps profiled."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("pmc_summary", os.path.join(ROOT, "tools", "pmc_summary.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rows = ["Dispatch_Id,Kernel_Name,Counter_Name,Counter_Value"]
    disp = 0
    for step in range(3):  # three steps: one k_meet3 and one k_meet4d launch each; k_bibfs only in the first
        for name, fetch_kb, write_kb in (("void pgq::k_meet3<false, false, 2>(long, long const*)", 400000.0, 6000.0),
                                         ("void pgq::k_meet4d<false, false>(pgq::MeetQueue)", 20000.0, 10000.0)):
            disp += 1
            rows += ["%d,\"%s\",FETCH_SIZE,%f" % (disp, name, fetch_kb), "%d,\"%s\",WRITE_SIZE,%f" % (disp, name, write_kb)]
        if step == 0:
            disp += 1
            rows += ["%d,\"void pgq::k_bibfs<false>(pgq::MeetQueue)\",FETCH_SIZE,50.0" % disp,
                     "%d,\"void pgq::k_bibfs<false>(pgq::MeetQueue)\",WRITE_SIZE,0.0" % disp]
    f = tmp_path / "p_counter_collection.csv"
    f.write_text("\n".join(rows) + "\n")
    out = mod.summarise([str(f)])
    meet, meet4 = (2 * 400000.0 + 6000.0) * 1024, (2 * 20000.0 + 10000.0) * 1024
    assert out["meet"]["kernels"] == ["k_meet3<false, false, 2>"] and out["meet"]["hbm_bytes_per_launch"] == meet
    assert out["meet4"]["hbm_bytes_per_launch"] == meet4 and out["bibfs"]["launches_profiled"] == 1
    chain = out["prepass_chain"]
    assert chain["steps_profiled"] == 3
    assert abs(chain["hbm_bytes_per_step"] - (meet + meet4 + 2 * 50.0 * 1024 / 3)) < 1.0


def test_bench_strided_sample_of_the_cross_product_covers_every_source():
    """bench.py's msbfs_cross leg compares a strided sample with the CPU port: every 256th row of the 2048 x 1024 product
    touches all 2048 sources (= all lanes of the batch); the first 8192 rows, round 3's sample, are 8 of them."""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    cp = bench.cross_pairs(448626, 2048 * 1024, 2048, bench.PAIR_SEED["snb_cross"])
    assert cp.shape == (2048 * 1024, 2) and len(np.unique(cp[:, 0])) == 2048
    sel = np.arange(8192) * (len(cp) // 8192)
    assert len(np.unique(cp[sel, 0])) == 2048
    assert len(np.unique(cp[:8192, 0])) == 8


def test_stats_struct_layout_matches_the_header(tmp_path):
    """pgq_stats_t grew at its end in round 5 (spec_batches ... host_waits): the ctypes mirror must have the header's size
    and field offsets, or every statistic after the first mismatch reads garbage."""
    import ctypes
    import subprocess
    from duckpgq_extension_amd import binding
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "pgq_hip.h"\nint main(void){printf("%zu %zu %zu %zu %zu\\n", '
                   'sizeof(pgq_stats_t), offsetof(pgq_stats_t, algo_bytes), offsetof(pgq_stats_t, launches), '
                   'offsetof(pgq_stats_t, spec_batches), offsetof(pgq_stats_t, host_waits));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)])
    size, o_algo, o_launch, o_spec, o_waits = (int(x) for x in subprocess.check_output([str(exe)]).split())
    S = binding.Stats
    assert ctypes.sizeof(S) == size
    assert (S.algo_bytes.offset, S.launches.offset, S.spec_batches.offset, S.host_waits.offset) == (o_algo, o_launch, o_spec, o_waits)
