"""Thin numpy-facing wrapper of one engine space (all arithmetic happens in libehx.so on the GPU).

Mirrors the embeddingstore service surface for one space
(embeddinghub/embeddingstore/embedding_store.proto:9-19): Set / MultiSet / Get / NearestNeighbor /
FreezeSpace, and the Go VectorStoreTable surface (provider/online.go:50-64): Set / Get / Nearest.
"""
import ctypes as C
import itertools

import numpy as np

from . import _lib
from ._lib import EhxError, Params, Stats, check

_counter = itertools.count()


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


class Space:
    def __init__(self, name, dims, metric=_lib.METRIC_L2SQ, mode=_lib.MODE_FLAT, M=0, ef_construction=0,
                 ef=0, seed=0, initial_capacity=0, build_batch=0, dtype=_lib.DTYPE_F32, scan=_lib.SCAN_AUTO, shards=0,
                 search_width=0, _handle=None):
        self._L = _lib.load()
        self.name, self.dims, self.metric = name, int(dims), metric
        self._M = M or 16
        if _handle is not None:
            self._h = _handle
            return
        p = Params(mode=mode, M=M, ef_construction=ef_construction, ef=ef, seed=seed,
                   initial_capacity=initial_capacity, build_batch=build_batch, scan=scan, shards=shards,
                   search_width=search_width)
        h = C.c_void_p()
        nm = name.encode()
        check(self._L.ehx_space_create(nm, len(nm), self.dims, metric, dtype, C.byref(p), C.byref(h)))
        self._h = h

    @classmethod
    def unique(cls, prefix, dims, **kw):
        return cls("%s-%d" % (prefix, next(_counter)), dims, **kw)

    @classmethod
    def open(cls, name):
        L = _lib.load()
        h = C.c_void_p()
        nm = name.encode()
        check(L.ehx_space_open(nm, len(nm), C.byref(h)))
        d = C.c_uint32()
        check(L.ehx_space_dims(h, C.byref(d)))
        return cls(name, d.value, _handle=h)

    def drop(self):
        if self._h:
            check(self._L.ehx_space_drop(self._h))
            self._h = None

    def freeze(self):
        check(self._L.ehx_space_freeze(self._h))

    def reserve(self, rows):
        check(self._L.ehx_space_reserve(self._h, rows))

    def __len__(self):
        n = C.c_uint64()
        check(self._L.ehx_space_size(self._h, C.byref(n)))
        return n.value

    # ---- writes ----
    def set(self, key, vec):
        v, pv = _f32(vec)
        if v.size != self.dims:
            raise ValueError("expected %d dims, got %d" % (self.dims, v.size))
        k = key.encode() if isinstance(key, str) else bytes(key)
        check(self._L.ehx_set(self._h, k, len(k), pv))

    def set_batch(self, keys, vecs):
        v, pv = _f32(vecs)
        v = v.reshape(-1, self.dims)
        n, arr, lens, keep = marshal_keys(keys)
        if n != v.shape[0]:
            raise ValueError("keys/vecs length mismatch")
        check(self._L.ehx_set_batch(self._h, n, arr, lens, pv))
        del keep

    def prepare_batch(self, keys, vecs):
        """Marshal a batch once (key arrays, contiguous fp32 rows) so that set_prepared() is ONE C call with no Python
        work in front of it — what a cgo / C++ caller's BatchSet looks like; used by the concurrency measurements,
        where a Python writer thread would otherwise hold the GIL for milliseconds per chunk."""
        v, pv = _f32(vecs)
        v = v.reshape(-1, self.dims)
        n, arr, lens, keep = marshal_keys(keys)
        if n != v.shape[0]:
            raise ValueError("keys/vecs length mismatch")
        return (n, arr, lens, pv, v, keep)

    def set_prepared(self, prep):
        check(self._L.ehx_set_batch(self._h, prep[0], prep[1], prep[2], prep[3]))

    def graph_import(self, level0, levels, upper, entry_point, max_level):
        """Attach an HNSW graph over the rows already Set (graph mode).

        level0: [n, 1+2M] u32 rows = (count, ids...); levels: [n] i32; upper: {(node, level>=1): ids}.
        """
        l0 = np.ascontiguousarray(level0, dtype=np.uint32)
        lv = np.ascontiguousarray(levels, dtype=np.int32)
        items = sorted(upper.items())
        un = np.array([key[0] for key, _ in items], dtype=np.uint32)
        ul = np.array([key[1] for key, _ in items], dtype=np.int32)
        off = np.zeros(len(items) + 1, dtype=np.uint64)
        if items:
            off[1:] = np.cumsum([len(v) for _, v in items])
            ids = np.concatenate([np.asarray(v, dtype=np.uint32) for _, v in items]).astype(np.uint32)
        else:
            ids = np.zeros(1, dtype=np.uint32)
        P = lambda a, t: a.ctypes.data_as(C.POINTER(t))  # noqa: E731
        check(self._L.ehx_graph_import(self._h, l0.shape[0], P(l0, C.c_uint32), P(lv, C.c_int32), len(items),
                                       P(un, C.c_uint32), P(ul, C.c_int32), P(off, C.c_uint64),
                                       P(ids, C.c_uint32), int(entry_point), int(max_level)))

    def graph_export(self):
        """-> level0 [n, 1+2M] u32, levels [n] i32, upper {(node, level): ids}, entry_point, max_level."""
        n = len(self)
        nl, ep, ml = C.c_uint64(), C.c_uint32(), C.c_int32()
        check(self._L.ehx_graph_export(self._h, None, None, None, None, 0, C.byref(nl), C.byref(ep), C.byref(ml)))
        M = self._M
        l0 = np.zeros((n, 1 + 2 * M), dtype=np.uint32)
        lv = np.zeros(n, dtype=np.int32)
        us = np.zeros(n, dtype=np.uint32)
        ul = np.zeros((max(nl.value, 1), M), dtype=np.uint32)
        P = lambda a, t: a.ctypes.data_as(C.POINTER(t))  # noqa: E731
        check(self._L.ehx_graph_export(self._h, P(l0, C.c_uint32), P(lv, C.c_int32), P(us, C.c_uint32),
                                       P(ul, C.c_uint32), ul.shape[0], C.byref(nl), C.byref(ep), C.byref(ml)))
        upper = {}
        for i in np.nonzero(lv > 0)[0]:
            for level in range(1, int(lv[i]) + 1):
                row = ul[int(us[i]) + level - 1]
                upper[(int(i), level)] = row[row != 0xFFFFFFFF].copy()
        return l0, lv, upper, ep.value, ml.value

    def set_ef(self, ef):
        check(self._L.ehx_space_set_ef(self._h, ef))

    def set_search_width(self, width):
        """graph spaces: level-0 expansions per search step (1 = the strict, hnswlib-order walk; 2 / 4 = the wide walk)"""
        check(self._L.ehx_space_set_search_width(self._h, width))

    def set_scan(self, scan):
        check(self._L.ehx_space_set_scan(self._h, scan))

    def scan_engine(self):
        """the engine that answers first right now: "f32" | "f16" | "i8" """
        e = C.c_uint32()
        check(self._L.ehx_space_scan_engine(self._h, C.byref(e)))
        return ("f32", "f16", "i8")[e.value]

    def fill_synthetic(self, seed, row0, n_rows, normalize):
        check(self._L.ehx_fill_synthetic(self._h, seed, row0, n_rows, int(bool(normalize))))

    def fill_manifold(self, seed, row0, n_rows, latent_dims, normalize):
        """EHX-MANIFOLD-1 rows (include/ehx_datagen.h) generated on the device: a latent_dims-dimensional linear subspace
        + 5 % noise"""
        check(self._L.ehx_fill_manifold(self._h, seed, row0, n_rows, latent_dims, int(bool(normalize))))

    # ---- reads ----
    def get(self, key):
        k = key.encode() if isinstance(key, str) else bytes(key)
        out = np.empty(self.dims, dtype=np.float32)
        check(self._L.ehx_get(self._h, k, len(k), out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def get_by_id(self, i):
        out = np.empty(self.dims, dtype=np.float32)
        check(self._L.ehx_get_by_id(self._h, int(i), out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def key_of(self, i):
        n = C.c_size_t()
        buf = C.create_string_buffer(4096)
        check(self._L.ehx_key_of(self._h, int(i), buf, 4096, C.byref(n)))
        return buf.raw[:n.value].decode()

    # ---- kNN ----
    def knn(self, queries, k):
        """-> ids [nq,k] u64, dist [nq,k] f32, count [nq] u32 (nearest first)."""
        q, pq = _f32(queries)
        q = q.reshape(-1, self.dims)
        nq = q.shape[0]
        ids = np.full((nq, max(k, 1)), np.uint64(2**64 - 1), dtype=np.uint64)
        dist = np.full((nq, max(k, 1)), np.inf, dtype=np.float32)
        cnt = np.zeros(nq, dtype=np.uint32)
        check(self._L.ehx_knn(self._h, nq, pq, k, ids.ctypes.data_as(C.POINTER(C.c_uint64)),
                              dist.ctypes.data_as(C.POINTER(C.c_float)),
                              cnt.ctypes.data_as(C.POINTER(C.c_uint32))))
        return ids[:, :k], dist[:, :k], cnt

    def knn_into(self, queries, k, ids, dist, cnt):
        """ehx_knn into the caller's arrays (C-contiguous: queries [nq, dims] f32, ids [nq, k] u64, dist [nq, k] f32,
        cnt [nq] u32) — nothing is allocated or converted on the way; callable from several threads at once."""
        nq = queries.shape[0]
        if (queries.dtype != np.float32 or ids.dtype != np.uint64 or dist.dtype != np.float32 or cnt.dtype != np.uint32
                or not (queries.flags.c_contiguous and ids.flags.c_contiguous and dist.flags.c_contiguous
                        and cnt.flags.c_contiguous)
                or queries.shape[1] != self.dims or ids.shape != (nq, k) or dist.shape != (nq, k) or cnt.shape != (nq,)):
            raise ValueError("knn_into: arrays of the wrong dtype / shape / layout")
        check(self._L.ehx_knn(self._h, nq, queries.ctypes.data_as(C.POINTER(C.c_float)), k,
                              ids.ctypes.data_as(C.POINTER(C.c_uint64)), dist.ctypes.data_as(C.POINTER(C.c_float)),
                              cnt.ctypes.data_as(C.POINTER(C.c_uint32))))

    def knn_keys(self, queries, k):
        """-> list (per query) of key lists, nearest first."""
        q, pq = _f32(queries)
        q = q.reshape(-1, self.dims)
        nq = q.shape[0]
        ids = np.zeros((nq, max(k, 1)), dtype=np.uint64)
        dist = np.zeros((nq, max(k, 1)), dtype=np.float32)
        cnt = np.zeros(nq, dtype=np.uint32)
        off = np.zeros(nq * k + 1, dtype=np.uint64)
        cap = 1 << 16
        while True:
            arena = C.create_string_buffer(cap)
            rc = self._L.ehx_knn_keys(self._h, nq, pq, k, ids.ctypes.data_as(C.POINTER(C.c_uint64)),
                                      dist.ctypes.data_as(C.POINTER(C.c_float)),
                                      cnt.ctypes.data_as(C.POINTER(C.c_uint32)), arena, cap,
                                      off.ctypes.data_as(C.POINTER(C.c_uint64)))
            if rc == _lib.ERANGE:
                cap *= 4
                continue
            check(rc)
            break
        raw = arena.raw
        out = []
        for i in range(nq):
            out.append([raw[int(off[i * k + j]):int(off[i * k + j + 1])].decode() for j in range(int(cnt[i]))])
        return out

    def knn_by_key(self, key, k):
        kb = key.encode() if isinstance(key, str) else bytes(key)
        ids = np.zeros(max(k, 1), dtype=np.uint64)
        dist = np.zeros(max(k, 1), dtype=np.float32)
        cnt = C.c_uint32()
        check(self._L.ehx_knn_by_key(self._h, kb, len(kb), k, ids.ctypes.data_as(C.POINTER(C.c_uint64)),
                                     dist.ctypes.data_as(C.POINTER(C.c_float)), C.byref(cnt)))
        return ids[:cnt.value], dist[:cnt.value]

    def knn_by_key_keys(self, key, k):
        """the NearestNeighbor RPC by key in ONE engine call -> list of neighbour keys, nearest first"""
        kb = key.encode() if isinstance(key, str) else bytes(key)
        cnt = C.c_uint32()
        off = np.zeros(k + 1, dtype=np.uint64)
        cap = 1 << 12
        while True:
            arena = C.create_string_buffer(cap)
            rc = self._L.ehx_knn_by_key_keys(self._h, kb, len(kb), k, None, None, C.byref(cnt), arena, cap,
                                             off.ctypes.data_as(C.POINTER(C.c_uint64)))
            if rc == _lib.ERANGE:
                cap *= 8
                continue
            check(rc)
            break
        raw = arena.raw
        o = off.tolist()
        return [raw[o[j]:o[j + 1]].decode() for j in range(cnt.value)]

    # ---- device-resident (torch tensors on the GPU) ----
    def knn_device(self, d_queries, k, d_ids, d_dist, d_count, stream=None):
        """All arguments are device pointers (ints) or torch CUDA tensors; enqueues on `stream`."""
        def ptr(t):
            return C.c_void_p(t.data_ptr() if hasattr(t, "data_ptr") else int(t))
        nq = d_queries.shape[0] if hasattr(d_queries, "shape") else None
        if nq is None:
            raise ValueError("pass torch tensors (queries [nq, dims])")
        check(self._L.ehx_knn_device(self._h, C.c_void_p(stream or 0), nq, ptr(d_queries), k, ptr(d_ids),
                                     ptr(d_dist), ptr(d_count)))

    def stats(self):
        st = Stats()
        check(self._L.ehx_stats(self._h, C.byref(st)))
        return {f: getattr(st, f) for f, _ in Stats._fields_}

    def stats_reset(self):
        check(self._L.ehx_stats_reset(self._h))

    def graph_counters(self):
        """(rows fetched, level-0 expansions, upper-level expansions, prefetch hits, ...) since the last reset"""
        out = (C.c_uint64 * 12)()
        check(self._L.ehx_graph_counters(self._h, out, 12))
        return tuple(int(v) for v in out)  # [4:] phase timers of -DEHX_GRAPH_PROFILE builds, else zeros


def marshal_keys(keys):
    """keys (str or bytes) -> (n, `const char* const*`, `const size_t*`, keep-alive) for ehx_set_batch: ONE joined
    byte buffer and two numpy arrays (pointers = buffer address + running offsets, lengths) instead of a ctypes object
    per key — 1.1 ms instead of 4.2 ms per 8192 keys, which is as long as the engine itself takes for such a chunk.
    The returned keep-alive tuple owns the memory the pointers refer to: hold it until the call has returned."""
    ks = [k.encode() if isinstance(k, str) else bytes(k) for k in keys]
    n = len(ks)
    lens = np.fromiter(map(len, ks), dtype=np.uint64, count=n)
    blob = b"".join(ks)
    buf = C.create_string_buffer(blob, len(blob) + 1)
    ptrs = np.zeros(n, dtype=np.uint64)
    if n:
        np.cumsum(lens[:-1], out=ptrs[1:])
        ptrs += np.uint64(C.addressof(buf))
    return (n, ptrs.ctypes.data_as(C.POINTER(C.c_char_p)), lens.ctypes.data_as(C.POINTER(C.c_size_t)),
            (buf, ptrs, lens))


def nearest_neighbor_rpc(space, num, key="", embedding=None):
    """NearestNeighbor RPC semantics of embeddinghub/embeddingstore/server.cc:172-210 over a Space.

    Returns (grpc_status_code, keys): 0 OK, 3 INVALID_ARGUMENT, 5 NOT_FOUND.
    """
    has_key = key != ""
    has_vec = embedding is not None and len(embedding) != 0
    if has_key and has_vec:
        return 3, []
    if not has_key and not has_vec:
        return 3, []
    try:
        if has_key:
            return 0, space.knn_by_key_keys(key, num)   # (one engine call: lookup, search k + 1, drop the key, keys back)
        return 0, space.knn_keys(np.asarray(embedding, dtype=np.float32), num)[0]
    except EhxError as e:
        if e.code == _lib.ENOTFOUND:
            return 5, []
        raise
