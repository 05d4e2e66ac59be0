"""ORACLE — TEST INFRASTRUCTURE ONLY.  ctypes binding of oracle/liboracle.so."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

METRIC_L2, METRIC_IP, METRIC_COSINE = 0, 1, 2


def build(force=False):
    """Compile liboracle.so with g++ (seconds)."""
    so = os.path.join(_HERE, "liboracle.so")
    if force or not os.path.exists(so):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"] + (["-B"] if force else []))
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = build()
        L = C.CDLL(so)
        f32p, u64p, u32p, i32p = (C.POINTER(C.c_float), C.POINTER(C.c_uint64),
                                  C.POINTER(C.c_uint32), C.POINTER(C.c_int32))
        L.orc_dist.restype = C.c_float
        L.orc_dist.argtypes = [C.c_int, f32p, f32p, C.c_size_t]
        L.orc_normalize.argtypes = [f32p, f32p, C.c_size_t]
        L.orc_hnsw_new.restype = C.c_void_p
        L.orc_hnsw_new.argtypes = [C.c_size_t, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t]
        L.orc_hnsw_free.argtypes = [C.c_void_p]
        L.orc_hnsw_add.restype = C.c_int
        L.orc_hnsw_add.argtypes = [C.c_void_p, f32p, C.c_uint64, C.c_char_p, C.c_size_t]
        L.orc_hnsw_add_rows.restype = C.c_double
        L.orc_hnsw_add_rows.argtypes = [C.c_void_p, f32p, C.c_size_t, C.c_uint64]
        L.orc_hnsw_add_rows_rounds.restype = C.c_double
        L.orc_hnsw_add_rows_rounds.argtypes = [C.c_void_p, f32p, C.c_size_t, C.c_uint64, C.c_size_t, C.c_size_t, C.c_int,
                                               C.c_char_p, C.c_size_t]
        L.orc_hnsw_add_rows_parallel.restype = C.c_double
        L.orc_hnsw_add_rows_parallel.argtypes = [C.c_void_p, f32p, C.c_size_t, C.c_uint64, C.c_int, C.c_char_p, C.c_size_t]
        L.orc_hnsw_resize.restype = C.c_int
        L.orc_hnsw_resize.argtypes = [C.c_void_p, C.c_size_t]
        L.orc_hnsw_set_ef.argtypes = [C.c_void_p, C.c_size_t]
        L.orc_hnsw_size.restype = C.c_size_t
        L.orc_hnsw_size.argtypes = [C.c_void_p]
        L.orc_hnsw_maxlevel.restype = C.c_int
        L.orc_hnsw_maxlevel.argtypes = [C.c_void_p]
        L.orc_hnsw_enterpoint.restype = C.c_uint32
        L.orc_hnsw_enterpoint.argtypes = [C.c_void_p]
        L.orc_hnsw_level_of.restype = C.c_int
        L.orc_hnsw_level_of.argtypes = [C.c_void_p, C.c_uint32]
        L.orc_hnsw_search.restype = C.c_size_t
        L.orc_hnsw_search.argtypes = [C.c_void_p, f32p, C.c_size_t, u64p, f32p]
        L.orc_hnsw_search_batch.restype = C.c_double
        L.orc_hnsw_search_batch.argtypes = [C.c_void_p, f32p, C.c_size_t, C.c_size_t, u64p, f32p,
                                            u32p, C.c_int, u64p]
        L.orc_hnsw_search_level0.restype = C.c_size_t
        L.orc_hnsw_search_level0.argtypes = [C.c_void_p, C.c_uint32, f32p, C.c_size_t, u32p, f32p, u64p]
        L.orc_hnsw_export_level0.argtypes = [C.c_void_p, u32p]
        L.orc_hnsw_export_levels.argtypes = [C.c_void_p, i32p]
        L.orc_hnsw_export_upper.restype = C.c_uint32
        L.orc_hnsw_export_upper.argtypes = [C.c_void_p, C.c_uint32, C.c_int, u32p]
        L.orc_hnsw_export_vectors.argtypes = [C.c_void_p, f32p]
        L.orc_exhaustive.restype = C.c_double
        L.orc_exhaustive.argtypes = [f32p, C.c_size_t, C.c_size_t, C.c_int, f32p, C.c_size_t,
                                     C.c_size_t, u64p, f32p, u32p, C.c_int]
        L.orc_ann_new.restype = C.c_void_p
        L.orc_ann_new.argtypes = [C.c_size_t, C.c_size_t, C.c_int]
        L.orc_ann_free.argtypes = [C.c_void_p]
        L.orc_ann_set.argtypes = [C.c_void_p, C.c_char_p, f32p]
        L.orc_ann_size.restype = C.c_size_t
        L.orc_ann_size.argtypes = [C.c_void_p]
        L.orc_ann_approx_nearest.restype = C.c_size_t
        L.orc_ann_approx_nearest.argtypes = [C.c_void_p, f32p, C.c_size_t, C.c_char_p, C.c_size_t,
                                             C.POINTER(C.c_size_t)]
        L.orc_ann_nearest_rpc.restype = C.c_int
        L.orc_ann_nearest_rpc.argtypes = [C.c_void_p, C.c_int, C.c_char_p, f32p, C.c_size_t,
                                          C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t),
                                          C.POINTER(C.c_size_t)]
        L.orc_minstd_first.argtypes = [C.c_uint32, u32p, C.c_size_t]
        L.orc_levels.argtypes = [C.c_uint32, C.c_size_t, i32p, C.c_size_t]
        L.orc_philox.argtypes = [u32p, u32p, u32p]
        L.orc_gen_rows.argtypes = [C.c_uint64, C.c_uint64, C.c_size_t, C.c_size_t, C.c_int, f32p, C.c_int]
        L.orc_gen_manifold_rows.argtypes = [C.c_uint64, C.c_uint64, C.c_size_t, C.c_size_t, C.c_uint32, C.c_int, f32p, C.c_int]
        _LIB = L
    return _LIB


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


def _ptr(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def dist(metric, a, b):
    a, pa = _f32(a)
    b, pb = _f32(b)
    return float(lib().orc_dist(metric, pa, pb, a.size))


def normalize(v):
    v, pv = _f32(v)
    out = np.empty_like(v)
    lib().orc_normalize(pv, _ptr(out, C.c_float), v.size)
    return out


def exhaustive(X, Q, k, metric=METRIC_L2, threads=None, return_time=False):
    """Exhaustive kNN, (dist, id) order.  Returns ids [nq,k] u64, dists [nq,k] f32, counts [nq]."""
    X, pX = _f32(X)
    Q, pQ = _f32(Q)
    Q = Q.reshape(-1, X.shape[1])
    nq = Q.shape[0]
    ids = np.zeros((nq, max(k, 1)), dtype=np.uint64)
    dists = np.zeros((nq, max(k, 1)), dtype=np.float32)
    counts = np.zeros(nq, dtype=np.uint32)
    sec = lib().orc_exhaustive(pX, X.shape[0], X.shape[1], metric, pQ, nq, k, _ptr(ids, C.c_uint64),
                               _ptr(dists, C.c_float), _ptr(counts, C.c_uint32),
                               threads or os.cpu_count() or 1)
    ids, dists = ids[:, :k], dists[:, :k]
    return (ids, dists, counts, sec) if return_time else (ids, dists, counts)


class Hnsw:
    """HierarchicalNSW restatement (defaults = what index.cc:14-15 gets)."""

    def __init__(self, dim, metric=METRIC_L2, max_elements=128, M=16, ef_construction=200, seed=100):
        self.dim, self.metric, self.M = dim, metric, M
        self._h = C.c_void_p(lib().orc_hnsw_new(dim, metric, max_elements, M, ef_construction, seed))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_hnsw_free(self._h)
            self._h = None

    def add(self, v, label):
        v, pv = _f32(v)
        assert v.size == self.dim
        err = C.create_string_buffer(256)
        if lib().orc_hnsw_add(self._h, pv, int(label), err, 256) != 0:
            raise RuntimeError(err.value.decode())

    def add_rows(self, X, first_label=0):
        X, pX = _f32(X)
        return lib().orc_hnsw_add_rows(self._h, pX, X.shape[0], first_label)

    def add_rows_parallel(self, X, first_label=0, threads=0):
        """hnswlib's multi-threaded add_items, for CPU BASELINES only: the graph depends on thread timing, so no parity
        test may use it (add_rows is the oracle).  Returns the seconds spent."""
        X, pX = _f32(X)
        err = C.create_string_buffer(256)
        sec = lib().orc_hnsw_add_rows_parallel(self._h, pX, X.shape[0], first_label, threads or os.cpu_count() or 1,
                                               err, 256)
        if sec < 0:
            raise RuntimeError(err.value.decode())
        return sec

    def add_rows_rounds(self, X, first_label=0, div=128, cap=4096, threads=0):
        """A CPU MODEL of the engine's bulk build, for studies only: rounds of graph_size / div rows (at most cap) that do
        not see each other, linked in ascending id order.  Returns the seconds spent."""
        X, pX = _f32(X)
        err = C.create_string_buffer(256)
        sec = lib().orc_hnsw_add_rows_rounds(self._h, pX, X.shape[0], first_label, div, cap,
                                             threads or os.cpu_count() or 1, err, 256)
        if sec < 0:
            raise RuntimeError(err.value.decode())
        return sec

    def resize(self, cap):
        if lib().orc_hnsw_resize(self._h, cap) != 0:
            raise RuntimeError("resize failed")

    def set_ef(self, ef):
        lib().orc_hnsw_set_ef(self._h, ef)

    def __len__(self):
        return lib().orc_hnsw_size(self._h)

    @property
    def maxlevel(self):
        return lib().orc_hnsw_maxlevel(self._h)

    @property
    def enterpoint(self):
        return lib().orc_hnsw_enterpoint(self._h)

    def search(self, q, k):
        q, pq = _f32(q)
        labels = np.zeros(max(k, 1), dtype=np.uint64)
        dists = np.zeros(max(k, 1), dtype=np.float32)
        c = lib().orc_hnsw_search(self._h, pq, k, _ptr(labels, C.c_uint64), _ptr(dists, C.c_float))
        return labels[:c], dists[:c]

    def search_batch(self, Q, k, threads=1):
        Q, pQ = _f32(Q)
        Q = Q.reshape(-1, self.dim)
        nq = Q.shape[0]
        labels = np.zeros((nq, max(k, 1)), dtype=np.uint64)
        dists = np.zeros((nq, max(k, 1)), dtype=np.float32)
        counts = np.zeros(nq, dtype=np.uint32)
        stats = np.zeros(5, dtype=np.uint64)
        sec = lib().orc_hnsw_search_batch(self._h, pQ, nq, k, _ptr(labels, C.c_uint64),
                                          _ptr(dists, C.c_float), _ptr(counts, C.c_uint32), threads,
                                          _ptr(stats, C.c_uint64))
        st = dict(zip(["n_dist", "n_hops0", "n_hops_up", "metric_hops", "metric_distance_computations"],
                      [int(x) for x in stats]))
        return labels[:, :k], dists[:, :k], counts, sec, st

    def search_level0(self, ep, q_prepared, ef):
        q, pq = _f32(q_prepared)
        ids = np.zeros(ef + 1, dtype=np.uint32)
        dists = np.zeros(ef + 1, dtype=np.float32)
        stats = np.zeros(2, dtype=np.uint64)
        c = lib().orc_hnsw_search_level0(self._h, ep, pq, ef, _ptr(ids, C.c_uint32),
                                         _ptr(dists, C.c_float), _ptr(stats, C.c_uint64))
        return ids[:c], dists[:c], int(stats[0]), int(stats[1])

    def export_graph(self):
        """level0 [n, 1+2M] u32 (count, ids), levels [n] i32, upper {(id, level): ids}."""
        n = len(self)
        l0 = np.zeros((n, 2 * self.M + 1), dtype=np.uint32)
        lv = np.zeros(n, dtype=np.int32)
        if n:
            lib().orc_hnsw_export_level0(self._h, _ptr(l0, C.c_uint32))
            lib().orc_hnsw_export_levels(self._h, _ptr(lv, C.c_int32))
        upper = {}
        buf = np.zeros(self.M, dtype=np.uint32)
        for i in np.nonzero(lv > 0)[0]:
            for level in range(1, int(lv[i]) + 1):
                c = lib().orc_hnsw_export_upper(self._h, int(i), level, _ptr(buf, C.c_uint32))
                upper[(int(i), level)] = buf[:c].copy()
        return l0, lv, upper

    def export_vectors(self):
        out = np.zeros((len(self), self.dim), dtype=np.float32)
        if len(self):
            lib().orc_hnsw_export_vectors(self._h, _ptr(out, C.c_float))
        return out


class AnnIndex:
    """embeddinghub/embeddingstore/index.{h,cc} ANNIndex + server.cc:172-210 RPC semantics."""

    def __init__(self, dims, init_cap=128, metric=METRIC_L2):
        self.dims = dims
        self._h = C.c_void_p(lib().orc_ann_new(dims, init_cap, metric))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_ann_free(self._h)
            self._h = None

    def set(self, key, value):
        v, pv = _f32(value)
        assert v.size == self.dims
        lib().orc_ann_set(self._h, key.encode(), pv)

    def size(self):
        return lib().orc_ann_size(self._h)

    @staticmethod
    def _unpack(buf, nbytes, nkeys):
        if nkeys == 0:
            return []
        return buf.raw[:nbytes - 1].decode().split("\n")

    def approx_nearest(self, value, num):
        v, pv = _f32(value)
        cap = 1 << 20
        buf = C.create_string_buffer(cap)
        nk = C.c_size_t(0)
        nb = lib().orc_ann_approx_nearest(self._h, pv, num, buf, cap, C.byref(nk))
        return self._unpack(buf, nb, nk.value)

    def nearest_neighbor_rpc(self, num, key="", embedding=None):
        """Returns (grpc_status_code, keys)."""
        emb = np.zeros(0, dtype=np.float32) if embedding is None else np.ascontiguousarray(embedding, dtype=np.float32)
        cap = 1 << 20
        buf = C.create_string_buffer(cap)
        nk, nb = C.c_size_t(0), C.c_size_t(0)
        st = lib().orc_ann_nearest_rpc(self._h, num, key.encode(), _ptr(emb, C.c_float), emb.size, buf,
                                       cap, C.byref(nk), C.byref(nb))
        return st, (self._unpack(buf, nb.value, nk.value) if st == 0 else [])


def minstd_first(seed, n):
    out = np.zeros(n, dtype=np.uint32)
    lib().orc_minstd_first(seed, _ptr(out, C.c_uint32), n)
    return out


def levels(seed, M, n):
    out = np.zeros(n, dtype=np.int32)
    lib().orc_levels(seed, M, _ptr(out, C.c_int32), n)
    return out


def philox(ctr, key):
    c = np.asarray(ctr, dtype=np.uint32)
    k = np.asarray(key, dtype=np.uint32)
    out = np.zeros(4, dtype=np.uint32)
    lib().orc_philox(_ptr(c, C.c_uint32), _ptr(k, C.c_uint32), _ptr(out, C.c_uint32))
    return out


def gen_rows(seed, row0, n_rows, dim, normalize=False, threads=None):
    out = np.zeros((n_rows, dim), dtype=np.float32)
    lib().orc_gen_rows(seed, row0, n_rows, dim, int(bool(normalize)), _ptr(out, C.c_float),
                       threads or os.cpu_count() or 1)
    return out


def gen_manifold_rows(seed, row0, n_rows, dim, latent, normalize=False, threads=None):
    """EHX-MANIFOLD-1 rows (include/ehx_datagen.h): points of a `latent`-dimensional linear subspace + 5 % noise"""
    assert 1 <= latent <= 64
    out = np.zeros((n_rows, dim), dtype=np.float32)
    lib().orc_gen_manifold_rows(seed, row0, n_rows, dim, latent, int(bool(normalize)), _ptr(out, C.c_float),
                                threads or os.cpu_count() or 1)
    return out
