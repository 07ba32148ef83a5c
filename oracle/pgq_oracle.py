"""ctypes binding of the CPU oracle (oracle/libpgq_oracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libpgq_oracle.so")


class Vec(C.Structure):
    """UnifiedVectorFormat stand-in: data + optional sel (u32) + optional validity (u64 words)."""
    _fields_ = [("data", C.c_void_p), ("sel", C.c_void_p), ("validity", C.c_void_p)]


def build(force=False):
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(
            os.path.join(_HERE, "pgq_oracle.cpp")):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"])


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        L.ora_last_error.restype = C.c_char_p
        L.ora_csr_new.restype = C.c_void_p
        L.ora_csr_free.argtypes = [C.c_void_p]
        for name in ("ora_csr_v", "ora_csr_e", "ora_csr_edge_ids", "ora_csr_w", "ora_csr_w_double", "ora_child"):
            getattr(L, name).restype = C.c_void_p
            getattr(L, name).argtypes = [C.c_void_p]
        for name in ("ora_csr_vsize", "ora_csr_esize", "ora_child_len"):
            getattr(L, name).restype = C.c_int64
            getattr(L, name).argtypes = [C.c_void_p]
        L.ora_csr_w_type.argtypes = [C.c_void_p]
        L.ora_create_csr_vertex.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ora_create_csr_edge.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, Vec, Vec, Vec, Vec,
                                          C.c_int, C.c_void_p, C.c_void_p]
        L.ora_csr_adopt.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_int]
        L.ora_iterativelength.argtypes = [C.c_void_p, C.c_int64, C.c_int64, Vec, Vec, C.c_void_p, C.c_void_p,
                                          C.c_void_p]
        L.ora_iterativelength2.argtypes = [C.c_void_p, C.c_int64, C.c_int64, Vec, Vec, C.c_void_p, C.c_void_p]
        L.ora_shortestpath.argtypes = [C.c_void_p, C.c_int64, C.c_int64, Vec, Vec, C.c_void_p, C.c_void_p,
                                       C.c_void_p]
        L.ora_cheapest_path_length.argtypes = [C.c_void_p, C.c_int64, C.c_int64, Vec, Vec, C.c_void_p, C.c_void_p]
        L.ora_lean_iterativelength.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                               C.c_void_p, C.c_void_p, C.c_int]
        L.ora_local_clustering_coefficient.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
        L.ora_pagerank.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
        L.ora_weakly_connected_component.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ora_lean_shortestpath.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_void_p]
        L.ora_lean_cheapest_path_length.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p,
                                                    C.c_void_p, C.c_void_p]
        L.ora_baseline_run.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_int, C.c_void_p]
        _lib = L
    return _lib


class OracleError(RuntimeError):
    pass


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def make_vec(data, sel=None, valid=None, keep=None):
    """data: int64/float64 array; sel: uint32 row->position; valid: bool per *position* (True = valid)."""
    d = np.ascontiguousarray(data)
    s = None if sel is None else np.ascontiguousarray(sel, dtype=np.uint32)
    m = None
    if valid is not None:
        m = pack_validity(np.asarray(valid, dtype=bool))
    if keep is not None:
        keep.extend([d, s, m])
    return Vec(_ptr(d), _ptr(s), _ptr(m))


def pack_validity(valid_bool):
    n = len(valid_bool)
    words = np.zeros((n + 63) // 64, dtype=np.uint64)
    idx = np.nonzero(valid_bool)[0]
    np.bitwise_or.at(words, idx // 64, np.uint64(1) << (idx % 64).astype(np.uint64))
    return words


def unpack_validity(words, n):
    i = np.arange(n)
    return ((words[i // 64] >> (i % 64).astype(np.uint64)) & np.uint64(1)).astype(bool)


class OracleCSR:
    """The reference's `class CSR` (compressed_sparse_row.hpp:25-47) built by the reference's UDF sequence."""

    def __init__(self):
        self.h = lib().ora_csr_new()

    def __del__(self):
        if getattr(self, "h", None):
            lib().ora_csr_free(self.h)
            self.h = None

    # -- construction through the UDF bodies --------------------------------
    def create_csr_vertex(self, V, dense_id, cnt):
        dense_id, cnt = _i64(dense_id), _i64(cnt)
        out = np.empty(len(cnt), dtype=np.int64)
        lib().ora_create_csr_vertex(self.h, V, len(cnt), _ptr(dense_id), _ptr(cnt), _ptr(out))
        return out

    def create_csr_edge(self, V, e_sum, e_count, src, dst, eid, w=None, valid=None):
        keep = []
        n = len(src)
        wtype = 0
        wv = Vec(None, None, None)
        if w is not None:
            w = np.asarray(w)
            if w.dtype.kind == "f":
                wtype, w = 2, np.ascontiguousarray(w, dtype=np.float64)
            else:
                wtype, w = 1, _i64(w)
            wv = make_vec(w, keep=keep)
        sv = make_vec(_i64(src), valid=valid, keep=keep)
        dv = make_vec(_i64(dst), keep=keep)
        ev = make_vec(_i64(eid), keep=keep)
        out = np.zeros(n, dtype=np.int32)
        ov = np.zeros((n + 63) // 64 + 1, dtype=np.uint64)
        rc = lib().ora_create_csr_edge(self.h, V, e_sum, e_count, n, sv, dv, ev, wv, wtype, _ptr(out), _ptr(ov))
        if rc != 0:
            raise OracleError(lib().ora_last_error().decode())
        return out, unpack_validity(ov, n)

    @classmethod
    def from_edges(cls, V, src, dst, eid=None, w=None, chunk=2048):
        """Replays the SQL of compressed_sparse_row.cpp:234-251 single-threaded: degrees -> create_csr_vertex,
        then create_csr_edge over the edge rows in table order, 2048 rows per call."""
        src, dst = _i64(src), _i64(dst)
        eid = np.arange(len(src), dtype=np.int64) if eid is None else _i64(eid)
        c = cls()
        cnt = np.bincount(src, minlength=V).astype(np.int64)
        e_sum = int(c.create_csr_vertex(V, np.arange(V, dtype=np.int64), cnt).sum()) if V > 0 else 0
        for lo in range(0, max(len(src), 1), chunk):
            sl = slice(lo, lo + chunk)
            if len(src) == 0:
                break
            c.create_csr_edge(V, e_sum, len(src), src[sl], dst[sl], eid[sl], None if w is None else np.asarray(w)[sl])
        return c

    @classmethod
    def adopt(cls, V, offsets, adj, edge_ids=None, w=None):
        c = cls()
        offsets, adj = _i64(offsets), _i64(adj)
        wtype, wp = 0, None
        if w is not None:
            w = np.asarray(w)
            wtype = 2 if w.dtype.kind == "f" else 1
            w = np.ascontiguousarray(w, dtype=np.float64 if wtype == 2 else np.int64)
        eids = None if edge_ids is None else _i64(edge_ids)
        lib().ora_csr_adopt(c.h, V, _ptr(offsets), len(offsets), _ptr(adj), _ptr(eids), _ptr(w), wtype)
        return c

    # -- views ---------------------------------------------------------------
    def _arr(self, fn, n, dtype=np.int64):
        p = getattr(lib(), fn)(self.h)
        if not p or n == 0:
            return np.zeros(0, dtype=dtype)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_int64 if dtype == np.int64 else C.c_double)),
                                     shape=(n,)).copy()

    @property
    def v(self):
        return self._arr("ora_csr_v", lib().ora_csr_vsize(self.h))

    @property
    def e(self):
        return self._arr("ora_csr_e", lib().ora_csr_esize(self.h))

    @property
    def edge_ids(self):
        return self._arr("ora_csr_edge_ids", lib().ora_csr_esize(self.h))

    @property
    def w(self):
        t = self.w_type
        if t == 1:
            return self._arr("ora_csr_w", lib().ora_csr_esize(self.h))
        if t == 2:
            return self._arr("ora_csr_w_double", lib().ora_csr_esize(self.h), np.float64)
        return None

    @property
    def w_type(self):
        return lib().ora_csr_w_type(self.h)

    # -- searches (literal restatement, one call = one DataChunk) -------------
    def _search_vecs(self, src, dst, src_valid, src_sel, dst_sel, keep):
        sv = make_vec(_i64(src), sel=src_sel, valid=src_valid, keep=keep)
        dv = make_vec(_i64(dst), sel=dst_sel, keep=keep)
        n = len(src_sel) if src_sel is not None else len(src)
        return sv, dv, n

    def iterativelength(self, V, src, dst, src_valid=None, src_sel=None, dst_sel=None, variant=1, stats=False):
        keep = []
        sv, dv, n = self._search_vecs(src, dst, src_valid, src_sel, dst_sel, keep)
        out = np.zeros(n, dtype=np.int64)
        ov = np.zeros((n + 63) // 64 + 1, dtype=np.uint64)
        st = np.zeros(4, dtype=np.int64)
        if variant == 1:
            rc = lib().ora_iterativelength(self.h, V, n, sv, dv, _ptr(out), _ptr(ov), _ptr(st) if stats else None)
        else:
            rc = lib().ora_iterativelength2(self.h, V, n, sv, dv, _ptr(out), _ptr(ov))
        if rc != 0:
            raise OracleError(lib().ora_last_error().decode())
        valid = unpack_validity(ov, n)
        return (out, valid, st) if stats else (out, valid)

    def shortestpath(self, V, src, dst, src_valid=None, src_sel=None, dst_sel=None):
        """Returns a python list: per row either None (NULL) or the [v,e,v,...] list."""
        keep = []
        sv, dv, n = self._search_vecs(src, dst, src_valid, src_sel, dst_sel, keep)
        off = np.zeros(n, dtype=np.uint64)
        ln = np.zeros(n, dtype=np.uint64)
        ov = np.zeros((n + 63) // 64 + 1, dtype=np.uint64)
        rc = lib().ora_shortestpath(self.h, V, n, sv, dv, _ptr(off), _ptr(ln), _ptr(ov))
        if rc != 0:
            raise OracleError(lib().ora_last_error().decode())
        return self._paths(off, ln, unpack_validity(ov, n))

    def _paths(self, off, ln, valid):
        child = self._arr("ora_child", lib().ora_child_len(self.h))
        return [child[int(o):int(o) + int(l)].tolist() if ok else None for o, l, ok in zip(off, ln, valid)]

    def cheapest_path_length(self, V, src, dst, src_valid=None, dst_valid=None):
        keep = []
        sv = make_vec(_i64(src), valid=src_valid, keep=keep)
        dv = make_vec(_i64(dst), valid=dst_valid, keep=keep)
        n = len(src)
        out = np.zeros(n, dtype=np.int64 if self.w_type == 1 else np.float64)
        ov = np.zeros((n + 63) // 64 + 1, dtype=np.uint64)
        rc = lib().ora_cheapest_path_length(self.h, V, n, sv, dv, _ptr(out), _ptr(ov))
        if rc != 0:
            raise OracleError(lib().ora_last_error().decode())
        return out, unpack_validity(ov, n)

    # -- lean per-pair equivalents --------------------------------------------
    def lean_iterativelength(self, V, src, dst, nthreads=1, with_te=False):
        src, dst = _i64(src), _i64(dst)
        n = len(src)
        out = np.zeros(n, dtype=np.int64)
        ov = np.zeros((n + 63) // 64 + 1, dtype=np.uint64)
        te = np.zeros(n, dtype=np.int64) if with_te else None
        lib().ora_lean_iterativelength(self.h, V, n, _ptr(src), _ptr(dst), _ptr(out), _ptr(ov), _ptr(te), nthreads)
        valid = unpack_validity(ov, n)
        return (out, valid, te) if with_te else (out, valid)

    def lean_shortestpath(self, V, src, dst):
        src, dst = _i64(src), _i64(dst)
        n = len(src)
        off = np.zeros(n, dtype=np.uint64)
        ln = np.zeros(n, dtype=np.uint64)
        ov = np.zeros((n + 63) // 64 + 1, dtype=np.uint64)
        lib().ora_lean_shortestpath(self.h, V, n, _ptr(src), _ptr(dst), _ptr(off), _ptr(ln), _ptr(ov))
        return self._paths(off, ln, unpack_validity(ov, n))

    def lean_cheapest_path_length(self, V, src, dst):
        src, dst = _i64(src), _i64(dst)
        n = len(src)
        out = np.zeros(n, dtype=np.int64 if self.w_type == 1 else np.float64)
        ov = np.zeros((n + 63) // 64 + 1, dtype=np.uint64)
        lib().ora_lean_cheapest_path_length(self.h, V, n, _ptr(src), _ptr(dst), _ptr(out), _ptr(ov))
        return out, unpack_validity(ov, n)

    # -- the other CSR consumers (literal restatements) -------------------------
    def local_clustering_coefficient(self, src):
        src = _i64(src)
        out = np.zeros(len(src), dtype=np.float32)
        lib().ora_local_clustering_coefficient(self.h, len(src), _ptr(src), _ptr(out))
        return out

    def pagerank(self):
        """(rank[V+2], iterations)"""
        n = lib().ora_csr_vsize(self.h)
        out = np.zeros(n, dtype=np.float64)
        it = C.c_int(0)
        lib().ora_pagerank(self.h, _ptr(out), C.byref(it))
        return out, it.value

    def weakly_connected_component(self, src):
        src = _i64(src)
        out = np.zeros(len(src), dtype=np.int64)
        ov = np.zeros((len(src) + 63) // 64 + 1, dtype=np.uint64)
        lib().ora_weakly_connected_component(self.h, len(src), _ptr(src), _ptr(out), _ptr(ov))
        return out, unpack_validity(ov, len(src))

    # -- cpu_baseline driver ---------------------------------------------------
    def baseline_run(self, which, V, src, dst, nthreads=1, stats=False):
        """which: 'iterativelength' | 'shortestpath'. DuckDB-style 2048-row chunks over nthreads workers."""
        src, dst = _i64(src), _i64(dst)
        n = len(src)
        out = np.zeros(n, dtype=np.int64)
        ov = np.zeros((n + 63) // 64 + 1, dtype=np.uint64)
        st = np.zeros(4, dtype=np.int64)
        lib().ora_baseline_run(self.h, 0 if which == "iterativelength" else 1, V, n, _ptr(src), _ptr(dst), _ptr(out),
                               _ptr(ov), nthreads, _ptr(st) if stats else None)
        valid = unpack_validity(ov, n)
        return (out, valid, st) if stats else (out, valid)
