"""ctypes binding of libpgq_hip.so / libpgq_udf.so (no compute here — see include/pgq_hip.h, include/pgq_udf.h)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
KCLASS_MAX = 16


class PgqError(RuntimeError):
    pass


class Vec(C.Structure):
    """pgq_vec_t: DuckDB UnifiedVectorFormat (data, optional selection, optional validity words)."""
    _fields_ = [("data", C.c_void_p), ("sel", C.c_void_p), ("validity", C.c_void_p)]


class Stats(C.Structure):
    _fields_ = [("batches", C.c_int64), ("levels", C.c_int64), ("push_levels", C.c_int64),
                ("pull_levels", C.c_int64), ("edges_scanned", C.c_int64), ("word_gathers", C.c_int64),
                ("frontier_vertices", C.c_int64), ("unique_sources", C.c_int64), ("pairs", C.c_int64),
                ("deferred_pairs", C.c_int64), ("meet_pairs", C.c_int64),
                ("algo_bytes", C.c_double * KCLASS_MAX), ("kernel_ms", C.c_double * KCLASS_MAX),
                ("launches", C.c_int64 * KCLASS_MAX), ("spec_batches", C.c_int64), ("spec_levels", C.c_int64),
                ("spec_aborts", C.c_int64), ("host_waits", C.c_int64), ("ball_segments", C.c_int64),
                ("ball_calls", C.c_int64)]


def lib_paths():
    # PGQ_HIP_LIB: another build of the device library (tuning sweeps compare compile-time variants); the product path is csrc/
    return os.environ.get("PGQ_HIP_LIB") or os.path.join(_CSRC, "libpgq_hip.so"), os.path.join(_CSRC, "libpgq_udf.so")


def build_native(force=False):
    """Compile csrc/ for gfx950 (hipcc cross-compiles without a GPU)."""
    args = ["make", "-C", _CSRC, "all"]
    if force:
        args.append("-B")
    subprocess.check_call(args)


_hip = None
_udf = None


def load_hip():
    global _hip
    if _hip is not None:
        return _hip
    path = lib_paths()[0]
    if not os.path.exists(path):
        raise PgqError("libpgq_hip.so is not built (%s): run `python -c 'import __graft_entry__ as g; g.build()'`"
                       % path)
    L = C.CDLL(path, mode=C.RTLD_GLOBAL)
    L.pgq_last_error.restype = C.c_char_p
    L.pgq_version.restype = C.c_char_p
    L.pgq_kclass_name.restype = C.c_char_p
    L.pgq_kclass_name.argtypes = [C.c_int]
    L.pgq_init.argtypes = [C.c_int]
    L.pgq_csr_upload.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                 C.POINTER(C.c_void_p)]
    L.pgq_csr_upload_device.argtypes = L.pgq_csr_upload.argtypes
    L.pgq_csr_upload_ex.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_uint,
                                    C.POINTER(C.c_void_p)]
    L.pgq_csr_build_device.argtypes = [C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                       C.POINTER(C.c_void_p)]
    L.pgq_csr_download.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.pgq_csr_free.argtypes = [C.c_void_p]
    for f in ("pgq_csr_num_vertices", "pgq_csr_num_edges", "pgq_csr_device_bytes"):
        getattr(L, f).restype = C.c_int64
        getattr(L, f).argtypes = [C.c_void_p]
    L.pgq_csr_w_type.argtypes = [C.c_void_p]
    L.pgq_iterativelength.argtypes = [C.c_void_p, C.c_int64, C.c_int64, Vec, Vec, C.c_void_p, C.c_void_p]
    L.pgq_shortestpath.argtypes = [C.c_void_p, C.c_int64, C.c_int64, Vec, Vec, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    L.pgq_cheapest_path_length.argtypes = [C.c_void_p, C.c_int64, C.c_int64, Vec, Vec, C.c_void_p, C.c_void_p]
    L.pgq_iterativelength_bulk_device.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
    L.pgq_traversed_edges_bulk_device.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.pgq_shortestpath_bulk_device.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                               C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
    L.pgq_cheapest_path_length_bulk_device.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                                       C.c_void_p]
    L.pgq_local_clustering_coefficient.argtypes = [C.c_void_p, C.c_int64, C.c_int64, Vec, C.c_void_p, C.c_void_p]
    L.pgq_local_clustering_coefficient_bulk_device.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
    L.pgq_pagerank.argtypes = [C.c_void_p, C.c_int64, C.c_int64, Vec, C.c_void_p, C.c_void_p]
    L.pgq_pagerank_device.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
    L.pgq_iterativelength_multi.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
    L.pgq_shortestpath_multi.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_int64, C.POINTER(C.c_int64)]
    L.pgq_cheapest_path_length_multi.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.pgq_init_devices.argtypes = [C.POINTER(C.c_int), C.c_int]
    L.pgq_init_mask.argtypes = [C.c_uint64]
    L.pgq_csr_replicate.argtypes = [C.c_void_p]
    L.pgq_set_option.argtypes = [C.c_char_p, C.c_char_p]
    L.pgq_get_option.argtypes = [C.c_char_p, C.POINTER(C.c_double)]
    L.pgq_get_default_option.argtypes = [C.c_char_p, C.POINTER(C.c_double)]
    L.pgq_csr_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
    L.pgq_csr_get_option.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_double)]
    L.pgq_iterativelength_bidirectional.argtypes = [C.c_void_p, C.c_int64, C.c_int64, Vec, Vec, C.c_void_p, C.c_void_p]
    L.pgq_weakly_connected_component.argtypes = [C.c_void_p, C.c_int64, C.c_int64, Vec, C.c_void_p, C.c_void_p]
    L.pgq_weakly_connected_component_device.argtypes = [C.c_void_p, C.c_void_p]
    L.pgq_get_stats.argtypes = [C.POINTER(Stats)]
    L.pgq_get_stats_sized.argtypes = [C.POINTER(Stats), C.c_size_t]
    L.pgq_measure_copy_bandwidth.argtypes = [C.c_int64, C.c_int, C.POINTER(C.c_double)]
    _hip = L
    return L


def load_udf():
    global _udf
    if _udf is not None:
        return _udf
    load_hip()
    path = lib_paths()[1]
    if not os.path.exists(path):
        raise PgqError("libpgq_udf.so is not built (%s)" % path)
    L = C.CDLL(path)
    L.pgq_udf_last_error.restype = C.c_char_p
    L.pgq_state_new.restype = C.c_void_p
    L.pgq_state_free.argtypes = [C.c_void_p]
    L.pgq_state_query_end.argtypes = [C.c_void_p]
    L.pgq_udf_create_csr_vertex.argtypes = [C.c_void_p, C.c_int32, C.c_int64, C.c_int64, Vec, Vec, C.c_void_p,
                                            C.c_void_p]
    L.pgq_udf_create_csr_edge.argtypes = [C.c_void_p, C.c_int32, C.c_int64, C.c_int64, C.c_int64, C.c_int64, Vec, Vec,
                                          Vec, C.POINTER(Vec), C.c_int, C.c_void_p, C.c_void_p]
    L.pgq_udf_bind_search.argtypes = [C.c_void_p, C.c_int32]
    for f in ("pgq_udf_iterativelength", "pgq_udf_iterativelength2", "pgq_udf_iterativelengthbidirectional",
              "pgq_udf_cheapest_path_length",
              "pgq_udf_reachability"):
        getattr(L, f).argtypes = [C.c_void_p, C.c_int32, C.c_int64, C.c_int64, Vec, Vec, C.c_void_p, C.c_void_p]
    L.pgq_udf_shortestpath.argtypes = [C.c_void_p, C.c_int32, C.c_int64, C.c_int64, Vec, Vec, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    L.pgq_udf_bind_cheapest.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int)]
    for f in ("pgq_udf_local_clustering_coefficient", "pgq_udf_pagerank", "pgq_udf_weakly_connected_component"):
        getattr(L, f).argtypes = [C.c_void_p, C.c_int32, C.c_int64, Vec, C.c_void_p, C.c_void_p]
    L.pgq_udf_delete_csr.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int)]
    L.pgq_udf_csr_get_w_type.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]
    for f in ("pgq_udf_scan_csr_v", "pgq_udf_scan_csr_e", "pgq_udf_scan_csr_w"):
        getattr(L, f).restype = C.c_int64
        getattr(L, f).argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64]
    L.pgq_udf_device_csr.restype = C.c_void_p
    L.pgq_udf_device_csr.argtypes = [C.c_void_p, C.c_int32]
    _udf = L
    return L


def _check(rc):
    if rc != 0:
        raise PgqError("pgq_hip error %d: %s" % (rc, load_hip().pgq_last_error().decode()))


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def pack_validity(valid):
    valid = np.asarray(valid, dtype=bool)
    words = np.zeros((len(valid) + 63) // 64, dtype=np.uint64)
    idx = np.nonzero(valid)[0]
    np.bitwise_or.at(words, idx // 64, np.uint64(1) << (idx % 64).astype(np.uint64))
    return words


def unpack_validity(words, n):
    i = np.arange(n)
    return ((words[i // 64] >> (i % 64).astype(np.uint64)) & np.uint64(1)).astype(bool)


def make_vec(data, sel=None, valid=None, keep=None):
    d = np.ascontiguousarray(data)
    s = None if sel is None else np.ascontiguousarray(sel, dtype=np.uint32)
    m = None if valid is None else pack_validity(valid)
    if keep is not None:
        keep.extend([d, s, m])
    return Vec(_p(d), _p(s), _p(m))


def set_option(key, value):
    _check(load_hip().pgq_set_option(str(key).encode(), str(value).encode()))


def get_default_option(key):
    """The value the library ships with (a process that sets nothing runs under it)."""
    v = C.c_double(0)
    _check(load_hip().pgq_get_default_option(str(key).encode(), C.byref(v)))
    return v.value


def get_option(key):
    v = C.c_double(0.0)
    _check(load_hip().pgq_get_option(str(key).encode(), C.byref(v)))
    return v.value


def kclass_names():
    L = load_hip()
    out = []
    for k in range(KCLASS_MAX):
        nm = L.pgq_kclass_name(k)
        if nm is None:
            break
        out.append(nm.decode())
    return out


def get_stats():
    st = Stats()
    _check(load_hip().pgq_get_stats_sized(C.byref(st), C.sizeof(st)))  # the library writes at most this mirror's size
    names = kclass_names()
    d = {f: getattr(st, f) for f, t in Stats._fields_ if t is C.c_int64}
    d["algo_bytes"] = {n: st.algo_bytes[i] for i, n in enumerate(names)}
    d["kernel_ms"] = {n: st.kernel_ms[i] for i, n in enumerate(names)}
    d["launches"] = {n: st.launches[i] for i, n in enumerate(names)}
    return d


def init_devices(devices):
    arr = (C.c_int * len(devices))(*devices)
    _check(load_hip().pgq_init_devices(arr, len(devices)))
    return load_hip().pgq_num_enabled_devices()


def reset_stats():
    _check(load_hip().pgq_reset_stats())


def copy_bandwidth_gbps(nbytes=1 << 30, iters=10):
    out = C.c_double(0)
    _check(load_hip().pgq_measure_copy_bandwidth(nbytes, iters, C.byref(out)))
    return out.value


def _lists(off, ln, valid, child):
    return [child[int(o):int(o) + int(l)].tolist() if ok else None for o, l, ok in zip(off, ln, valid)]


class DeviceCSR:
    """pgq_csr_t: a CSR resident in HBM (include/pgq_hip.h)."""

    def __init__(self, V, offsets, adj, edge_ids=None, w=None, handle=None, lazy_edge_ids=False):
        self.L = load_hip()
        self.V = int(V)
        self.h = C.c_void_p(handle) if handle else None
        self._owned = handle is None
        if handle is None:
            _check(self.L.pgq_init(-1))
            offsets, adj = _i64(offsets), _i64(adj)
            eids = None if edge_ids is None else _i64(edge_ids)
            wtype = 0
            if w is not None:
                w = np.asarray(w)
                wtype = 2 if w.dtype.kind == "f" else 1
                w = np.ascontiguousarray(w, dtype=np.float64 if wtype == 2 else np.int64)
            if len(adj) == 0:
                adj = np.zeros(1, dtype=np.int64)
            h = C.c_void_p()
            if lazy_edge_ids:  # PGQ_UPLOAD_LAZY_EDGE_IDS: the array must outlive the handle — this object keeps it
                self._keep_eids = eids
                _check(self.L.pgq_csr_upload_ex(self.V, _p(offsets), _p(adj), _p(eids), _p(w), wtype, 1, C.byref(h)))
            else:
                _check(self.L.pgq_csr_upload(self.V, _p(offsets), _p(adj), _p(eids), _p(w), wtype, C.byref(h)))
            self.h = h

    @classmethod
    def from_device_ptrs(cls, V, d_offsets, d_adj, d_edge_ids=0, d_w=0, w_type=0):
        """Arrays already in HBM (e.g. torch tensors' data_ptr())."""
        self = cls.__new__(cls)
        self.L = load_hip()
        self.V = int(V)
        self._owned = True
        _check(self.L.pgq_init(-1))
        h = C.c_void_p()
        _check(self.L.pgq_csr_upload_device(self.V, C.c_void_p(d_offsets), C.c_void_p(d_adj),
                                            C.c_void_p(d_edge_ids or None), C.c_void_p(d_w or None), w_type,
                                            C.byref(h)))
        self.h = h
        return self

    @classmethod
    def build_from_device_rows(cls, V, n_rows, d_src, d_dst, d_edge_id=0, d_w=0, w_type=0):
        """create_csr_vertex/create_csr_edge on the GPU: edge-table rows (device pointers) -> device CSR."""
        self = cls.__new__(cls)
        self.L = load_hip()
        self.V = int(V)
        self._owned = True
        _check(self.L.pgq_init(-1))
        h = C.c_void_p()
        _check(self.L.pgq_csr_build_device(self.V, n_rows, C.c_void_p(d_src), C.c_void_p(d_dst),
                                           C.c_void_p(d_edge_id or None), C.c_void_p(d_w or None), w_type, C.byref(h)))
        self.h = h
        return self

    def download(self):
        """(offsets[V+1], adj[E], edge_ids[E], w or None) as numpy, reference layout."""
        E = self.num_edges
        off = np.zeros(self.V + 1, dtype=np.int64)
        adj = np.zeros(max(E, 1), dtype=np.int64)
        eid = np.zeros(max(E, 1), dtype=np.int64)
        wt = self.w_type
        w = None if wt == 0 else np.zeros(max(E, 1), dtype=np.int64 if wt == 1 else np.float64)
        _check(self.L.pgq_csr_download(self.h, _p(off), _p(adj), _p(eid), _p(w)))
        return off, adj[:E], eid[:E], (None if w is None else w[:E])

    def close(self):
        if getattr(self, "h", None) and self._owned:
            self.L.pgq_csr_free(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def num_edges(self):
        return self.L.pgq_csr_num_edges(self.h)

    @property
    def w_type(self):
        return self.L.pgq_csr_w_type(self.h)

    @property
    def device_bytes(self):
        return self.L.pgq_csr_device_bytes(self.h)

    # -- chunk form (host arrays) ----------------------------------------------
    def _vecs(self, src, dst, src_valid, src_sel, dst_sel, dst_valid, keep):
        sv = make_vec(_i64(src), sel=src_sel, valid=src_valid, keep=keep)
        dv = make_vec(_i64(dst), sel=dst_sel, valid=dst_valid, keep=keep)
        n = len(src_sel) if src_sel is not None else len(src)
        return sv, dv, n

    def iterativelength(self, src, dst, src_valid=None, src_sel=None, dst_sel=None):
        keep = []
        sv, dv, n = self._vecs(src, dst, src_valid, src_sel, dst_sel, None, keep)
        out = np.zeros(n, dtype=np.int64)
        ov = np.zeros((n + 63) // 64 + 1, dtype=np.uint64)
        _check(self.L.pgq_iterativelength(self.h, self.V, n, sv, dv, _p(out), _p(ov)))
        return out, unpack_validity(ov, n)

    def shortestpath(self, src, dst, src_valid=None, src_sel=None, dst_sel=None, raw=False):
        keep = []
        sv, dv, n = self._vecs(src, dst, src_valid, src_sel, dst_sel, None, keep)
        off = np.zeros(n, dtype=np.uint64)
        ln = np.zeros(n, dtype=np.uint64)
        ov = np.zeros((n + 63) // 64 + 1, dtype=np.uint64)
        child = C.c_void_p()
        clen = C.c_uint64()
        _check(self.L.pgq_shortestpath(self.h, self.V, n, sv, dv, _p(off), _p(ln), _p(ov), C.byref(child),
                                       C.byref(clen)))
        ch = np.zeros(0, dtype=np.int64)
        if clen.value:
            ch = np.ctypeslib.as_array(C.cast(child, C.POINTER(C.c_int64)), shape=(clen.value,)).copy()
        if raw:  # the LIST vector as DuckDB sees it: list_entry_t{offset,length} per row, validity words, child payload
            return off, ln, ov, ch
        return _lists(off, ln, unpack_validity(ov, n), ch)

    def cheapest_path_length(self, src, dst, src_valid=None, dst_valid=None):
        keep = []
        sv, dv, n = self._vecs(src, dst, src_valid, None, None, dst_valid, keep)
        out = np.zeros(n, dtype=np.int64 if self.w_type == 1 else np.float64)
        ov = np.zeros((n + 63) // 64 + 1, dtype=np.uint64)
        _check(self.L.pgq_cheapest_path_length(self.h, self.V, n, sv, dv, _p(out), _p(ov)))
        return out, unpack_validity(ov, n)

    # -- bulk form (device pointers; torch tensors supply them) ------------------
    def iterativelength_bulk_ptr(self, n, d_src, d_dst, d_out):
        _check(self.L.pgq_iterativelength_bulk_device(self.h, n, C.c_void_p(d_src), C.c_void_p(d_dst),
                                                      C.c_void_p(d_out)))

    def local_clustering_coefficient(self, src, src_valid=None):
        keep = []
        sv = make_vec(_i64(src), valid=src_valid, keep=keep)
        n = len(src)
        out = np.zeros(n, dtype=np.float32)
        ov = np.zeros((n + 63) // 64 + 1, dtype=np.uint64)
        _check(self.L.pgq_local_clustering_coefficient(self.h, self.V, n, sv, _p(out), _p(ov)))
        return out, unpack_validity(ov, n)

    def pagerank(self, src):
        keep = []
        sv = make_vec(_i64(src), keep=keep)
        n = len(src)
        out = np.zeros(n, dtype=np.float64)
        ov = np.zeros((n + 63) // 64 + 1, dtype=np.uint64)
        _check(self.L.pgq_pagerank(self.h, self.V, n, sv, _p(out), _p(ov)))
        it = C.c_int(0)
        _check(self.L.pgq_pagerank_device(self.h, None, C.byref(it)))
        return out, unpack_validity(ov, n), it.value

    def iterativelength_multi(self, src, dst):
        """Host arrays answered by every enabled device (pgq_init_devices): shards + in-library gather."""
        src, dst = _i64(src), _i64(dst)
        out = np.zeros(len(src), dtype=np.int64)
        _check(self.L.pgq_iterativelength_multi(self.h, len(src), _p(src), _p(dst), _p(out)))
        return out

    def shortestpath_multi(self, src, dst):
        """Paths of host rows by every enabled device; returns (lengths, offsets, child) like the bulk form."""
        src, dst = _i64(src), _i64(dst)
        n = len(src)
        ln, off = np.zeros(n, dtype=np.int64), np.zeros(n, dtype=np.int64)
        used = C.c_int64(0)
        cap = max(1024, 16 * n)
        for _ in range(2):
            child = np.zeros(cap, dtype=np.int64)
            rc = self.L.pgq_shortestpath_multi(self.h, n, _p(src), _p(dst), _p(ln), _p(off), _p(child), cap, C.byref(used))
            if rc == 0 or used.value <= cap:
                break
            cap = used.value  # too small: the call reported what it needs
        _check(rc)
        return ln, off, child[:used.value]

    def cheapest_path_length_multi(self, src, dst):
        src, dst = _i64(src), _i64(dst)
        n = len(src)
        out = np.zeros(n, dtype=np.int64 if self.w_type == 1 else np.float64)
        ok = np.zeros(n, dtype=np.uint8)
        _check(self.L.pgq_cheapest_path_length_multi(self.h, n, _p(src), _p(dst), _p(out), _p(ok)))
        return out, ok.astype(bool)

    def replicate(self):
        _check(self.L.pgq_csr_replicate(self.h))

    def set_option(self, key, value):
        """An option of THIS handle only (pgq_csr_set_option); the process-wide set stays as it is."""
        _check(self.L.pgq_csr_set_option(self.h, str(key).encode(), str(value).encode()))

    def get_option(self, key):
        v = C.c_double(0.0)
        _check(self.L.pgq_csr_get_option(self.h, str(key).encode(), C.byref(v)))
        return v.value

    def traversed_edges_bulk_ptr(self, n, d_src, d_dst, d_out_len, d_out_te):
        _check(self.L.pgq_traversed_edges_bulk_device(self.h, n, C.c_void_p(d_src), C.c_void_p(d_dst),
                                                      C.c_void_p(d_out_len), C.c_void_p(d_out_te)))

    def shortestpath_bulk_ptr(self, n, d_src, d_dst, d_out_len, d_out_off, d_child, child_cap):
        used = C.c_int64(0)
        rc = self.L.pgq_shortestpath_bulk_device(self.h, n, C.c_void_p(d_src), C.c_void_p(d_dst), C.c_void_p(d_out_len),
                                                 C.c_void_p(d_out_off), C.c_void_p(d_child), child_cap, C.byref(used))
        return rc, used.value

    def cheapest_bulk_ptr(self, n, d_src, d_dst, d_out, d_ok):
        _check(self.L.pgq_cheapest_path_length_bulk_device(self.h, n, C.c_void_p(d_src), C.c_void_p(d_dst),
                                                           C.c_void_p(d_out), C.c_void_p(d_ok)))


class PgqState:
    """Host mirror of DuckPGQState + the scalar UDFs (include/pgq_udf.h); one call == one DataChunk."""

    def __init__(self):
        self.U = load_udf()
        self.s = C.c_void_p(self.U.pgq_state_new())

    def __del__(self):
        if getattr(self, "s", None):
            self.U.pgq_state_free(self.s)
            self.s = None

    def _ck(self, rc):
        if rc != 0:
            raise PgqError(self.U.pgq_udf_last_error().decode())

    def query_end(self):
        self._ck(self.U.pgq_state_query_end(self.s))

    def create_csr_vertex(self, csr_id, V, dense_id, cnt):
        keep = []
        n = len(dense_id)
        out = np.zeros(n, dtype=np.int64)
        ov = np.zeros((n + 63) // 64 + 1, dtype=np.uint64)
        self._ck(self.U.pgq_udf_create_csr_vertex(self.s, csr_id, V, n, make_vec(_i64(dense_id), keep=keep),
                                                  make_vec(_i64(cnt), keep=keep), _p(out), _p(ov)))
        return out

    def create_csr_edge(self, csr_id, V, e_sum, e_count, src, dst, eid, w=None, valid=None):
        keep = []
        n = len(src)
        out = np.zeros(n, dtype=np.int32)
        ov = np.zeros((n + 63) // 64 + 1, dtype=np.uint64)
        wv, wtype = None, 0
        if w is not None:
            w = np.asarray(w)
            wtype = 2 if w.dtype.kind == "f" else 1
            wvec = make_vec(np.ascontiguousarray(w, dtype=np.float64 if wtype == 2 else np.int64), keep=keep)
            wv = C.pointer(wvec)
            keep.append(wvec)
        self._ck(self.U.pgq_udf_create_csr_edge(self.s, csr_id, V, e_sum, e_count, n,
                                                make_vec(_i64(src), valid=valid, keep=keep),
                                                make_vec(_i64(dst), keep=keep), make_vec(_i64(eid), keep=keep), wv,
                                                wtype, _p(out), _p(ov)))
        return out, unpack_validity(ov, n)

    def build_csr(self, csr_id, V, src, dst, eid=None, w=None, chunk=2048):
        """The cte1 SQL of compressed_sparse_row.cpp:234-251, single-threaded: vertex degrees, then edge chunks."""
        src, dst = _i64(src), _i64(dst)
        eid = np.arange(len(src), dtype=np.int64) if eid is None else _i64(eid)
        cnt = np.bincount(src, minlength=V).astype(np.int64) if len(src) else np.zeros(V, dtype=np.int64)
        e_sum = 0
        for lo in range(0, V, chunk):
            e_sum += int(self.create_csr_vertex(csr_id, V, np.arange(lo, min(V, lo + chunk)), cnt[lo:lo + chunk]).sum())
        for lo in range(0, len(src), chunk):
            sl = slice(lo, lo + chunk)
            self.create_csr_edge(csr_id, V, e_sum, len(src), src[sl], dst[sl], eid[sl],
                                 None if w is None else np.asarray(w)[sl])
        return e_sum

    def bind_search(self, csr_id):
        self._ck(self.U.pgq_udf_bind_search(self.s, csr_id))

    def _search(self, fn, csr_id, V, src, dst, src_valid, src_sel, dst_sel, dst_valid, out):
        keep = []
        sv = make_vec(_i64(src), sel=src_sel, valid=src_valid, keep=keep)
        dv = make_vec(_i64(dst), sel=dst_sel, valid=dst_valid, keep=keep)
        n = len(src_sel) if src_sel is not None else len(src)
        ov = np.zeros((n + 63) // 64 + 1, dtype=np.uint64)
        self._ck(fn(self.s, csr_id, V, n, sv, dv, _p(out), _p(ov)))
        return unpack_validity(ov, n)

    def iterativelength(self, csr_id, V, src, dst, src_valid=None, src_sel=None, dst_sel=None, variant=1):
        n = len(src_sel) if src_sel is not None else len(src)
        out = np.zeros(n, dtype=np.int64)
        fn = {1: self.U.pgq_udf_iterativelength, 2: self.U.pgq_udf_iterativelength2,
              3: self.U.pgq_udf_iterativelengthbidirectional}[variant]
        ok = self._search(fn, csr_id, V, src, dst, src_valid, src_sel, dst_sel, None, out)
        return out, ok

    def reachability(self, csr_id, V, src, dst, src_valid=None):
        out = np.zeros(len(src), dtype=np.uint8)
        ok = self._search(self.U.pgq_udf_reachability, csr_id, V, src, dst, src_valid, None, None, None, out)
        return out.astype(bool), ok

    def shortestpath(self, csr_id, V, src, dst, src_valid=None, src_sel=None, dst_sel=None):
        keep = []
        sv = make_vec(_i64(src), sel=src_sel, valid=src_valid, keep=keep)
        dv = make_vec(_i64(dst), sel=dst_sel, keep=keep)
        n = len(src_sel) if src_sel is not None else len(src)
        off = np.zeros(n, dtype=np.uint64)
        ln = np.zeros(n, dtype=np.uint64)
        ov = np.zeros((n + 63) // 64 + 1, dtype=np.uint64)
        child = C.c_void_p()
        clen = C.c_uint64()
        self._ck(self.U.pgq_udf_shortestpath(self.s, csr_id, V, n, sv, dv, _p(off), _p(ln), _p(ov), C.byref(child),
                                             C.byref(clen)))
        ch = np.zeros(0, dtype=np.int64)
        if clen.value:
            ch = np.ctypeslib.as_array(C.cast(child, C.POINTER(C.c_int64)), shape=(clen.value,)).copy()
        return _lists(off, ln, unpack_validity(ov, n), ch)

    def bind_cheapest(self, csr_id):
        t = C.c_int(0)
        self._ck(self.U.pgq_udf_bind_cheapest(self.s, csr_id, C.byref(t)))
        return t.value

    def cheapest_path_length(self, csr_id, V, src, dst, src_valid=None, dst_valid=None):
        rt = self.bind_cheapest(csr_id)
        out = np.zeros(len(src), dtype=np.int64 if rt == 1 else np.float64)
        ok = self._search(self.U.pgq_udf_cheapest_path_length, csr_id, V, src, dst, src_valid, None, None, dst_valid,
                          out)
        return out, ok

    def _analytics(self, fn, csr_id, src, dtype):
        keep = []
        sv = make_vec(_i64(src), keep=keep)
        n = len(src)
        out = np.zeros(n, dtype=dtype)
        ov = np.zeros((n + 63) // 64 + 1, dtype=np.uint64)
        self._ck(fn(self.s, csr_id, n, sv, _p(out), _p(ov)))
        return out, unpack_validity(ov, n)

    def local_clustering_coefficient(self, csr_id, src):
        return self._analytics(self.U.pgq_udf_local_clustering_coefficient, csr_id, src, np.float32)

    def pagerank(self, csr_id, src):
        return self._analytics(self.U.pgq_udf_pagerank, csr_id, src, np.float64)

    def weakly_connected_component(self, csr_id, src):
        return self._analytics(self.U.pgq_udf_weakly_connected_component, csr_id, src, np.int64)

    def delete_csr(self, csr_id):
        f = C.c_int(0)
        self._ck(self.U.pgq_udf_delete_csr(self.s, csr_id, C.byref(f)))
        return bool(f.value)

    def csr_get_w_type(self, csr_id):
        t = C.c_int32(0)
        self._ck(self.U.pgq_udf_csr_get_w_type(self.s, csr_id, C.byref(t)))
        return t.value

    def get_csr_v(self, csr_id):
        n = self.U.pgq_udf_scan_csr_v(self.s, csr_id, None, 0)
        if n < 0:
            raise PgqError(self.U.pgq_udf_last_error().decode())
        out = np.zeros(n, dtype=np.int64)
        self.U.pgq_udf_scan_csr_v(self.s, csr_id, _p(out), n)
        return out

    def get_csr_e(self, csr_id):
        n = self.U.pgq_udf_scan_csr_e(self.s, csr_id, None, 0)
        if n < 0:
            raise PgqError(self.U.pgq_udf_last_error().decode())
        out = np.zeros(max(n, 1), dtype=np.int64)
        self.U.pgq_udf_scan_csr_e(self.s, csr_id, _p(out), n)
        return out[:n]

    def get_csr_w(self, csr_id):
        """get_csr_w (pgq_scan.cpp:113-153): int64 or float64 by the CSR's weight type."""
        n = self.U.pgq_udf_scan_csr_w(self.s, csr_id, None, 0)
        if n < 0:
            raise PgqError(self.U.pgq_udf_last_error().decode())
        out = np.zeros(max(n, 1), dtype=np.float64 if self.csr_get_w_type(csr_id) == 2 else np.int64)
        self.U.pgq_udf_scan_csr_w(self.s, csr_id, _p(out), n)
        return out[:n]

    def device_csr(self, csr_id):
        h = self.U.pgq_udf_device_csr(self.s, csr_id)
        if not h:
            raise PgqError(self.U.pgq_udf_last_error().decode())
        V = load_hip().pgq_csr_num_vertices(C.c_void_p(h))
        return DeviceCSR(V, None, None, handle=h)
