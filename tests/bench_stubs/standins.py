"""Stand-ins for the engine and torch.cuda so that bench.py's control flow runs on a CPU-only box: every kNN is answered
by the ORACLE's exhaustive scan (test infrastructure: tests may use the oracle, the product never does).

Used in-process by tests/test_bench_flow.py (pytest's monkeypatch applies them) and in CHILD processes by
tests/bench_stubs/sitecustomize.py (EHX_BENCH_STANDINS=1 in the environment: `python bench.py --gpus 2` then launches
two real ranks under torch.distributed.run whose process group is gloo and whose engine is this file)."""
import ctypes as C
import os
import sys
import types

import numpy as np
import torch

from oracle import pyoracle


class FakeSpace:
    spaces = {}
    _wmu = __import__("threading").Lock()

    def __init__(self, name, dims, metric=0, mode=0, initial_capacity=0, shards=0, dtype=0, build_batch=0, **kw):
        self.name, self.dims, self.metric, self.mode = name, dims, metric, mode
        self.X = np.zeros((0, dims), dtype=np.float32)
        self.keys = []
        self.half = dtype == 1     # an fp16-row space holds its rows rounded to binary16
        self.ef, self._scan, self._st = 10, 0, self._zero()
        FakeSpace.spaces[name] = self

    @staticmethod
    def _zero():
        return {"scan_ms_mean": 1.0, "scan_launches": 0, "n_uncertified": 0, "n_i8_queries": 0, "n_i8_fallback": 0,
                "n_filter_fallback": 0, "n_exhaustive": 0, "n_dist": 0, "n_hops": 0, "bytes_algorithmic": 0,
                "n_queries": 0, "last_scan_ms": 1.0}

    def _om(self):
        return {0: pyoracle.METRIC_L2, 1: pyoracle.METRIC_IP, 2: pyoracle.METRIC_COSINE}[self.metric]

    def fill_synthetic(self, seed, row0, n, normalize):
        rows = pyoracle.gen_rows(seed, row0, n, self.dims, normalize=bool(normalize))
        if self.half:
            rows = rows.astype(np.float16).astype(np.float32)
        self.keys += [str(i) for i in range(self.X.shape[0], self.X.shape[0] + n)]
        self.X = np.concatenate([self.X, rows])

    def fill_manifold(self, seed, row0, n, latent, normalize):
        rows = pyoracle.gen_manifold_rows(seed, row0, n, self.dims, latent, normalize=bool(normalize))
        self.X = np.concatenate([self.X, rows])

    def set_batch(self, keys, X):
        X = np.asarray(X, dtype=np.float32)
        if self.half:
            X = X.astype(np.float16).astype(np.float32)
        with FakeSpace._wmu:      # (the engine serialises writers of a space: ehx_space::wmu)
            self.keys += [k.decode() if isinstance(k, bytes) else str(k) for k in keys]
            self.X = np.concatenate([self.X, X])

    def key_of(self, i):
        return self.keys[i]

    def prepare_batch(self, keys, X):
        return keys, np.asarray(X, dtype=np.float32)

    def set_prepared(self, prep):
        self.set_batch(*prep)

    def graph_import(self, level0, levels, upper, entry_point, max_level):
        # an imported graph is walked as the oracle walks it: the stand-in rebuilds the oracle's (deterministic) graph
        self._hnsw = pyoracle.Hnsw(self.dims, self._om(), self.X.shape[0])
        self._hnsw.add_rows(self.X)
        assert self._hnsw.enterpoint == entry_point and self._hnsw.maxlevel == max_level

    def knn(self, Q, k):
        Q = np.asarray(Q, dtype=np.float32).reshape(-1, self.dims)
        if getattr(self, "_hnsw", None) is not None:
            self._hnsw.set_ef(self.ef)
            ids, dist, cnt, _, _ = self._hnsw.search_batch(Q, k, threads=1)
            return ids, dist, cnt
        ids, dist, cnt = pyoracle.exhaustive(self.X, Q, k, self._om())
        self._st["scan_launches"] += 1
        self._st["n_queries"] += Q.shape[0]
        self._st["n_dist"] += 100 * Q.shape[0]
        self._st["n_hops"] += 5 * Q.shape[0]
        self._st["bytes_algorithmic"] += 100 * Q.shape[0] * self.dims * 4
        self._st["n_i8_queries"] += Q.shape[0]
        return ids, dist, cnt

    def knn_into(self, Q, k, ids, dist, cnt):
        i, d, c = self.knn(Q, k)
        ids[...] = i
        dist[...] = d
        cnt[...] = c

    def knn_device(self, q, k, ids, dst, cnt, stream=None):
        i, d, c = self.knn(q.numpy(), k)
        ids.copy_(torch.from_numpy(i.astype(np.int64)))
        dst.copy_(torch.from_numpy(d))
        cnt.copy_(torch.from_numpy(c.astype(np.int32)))

    def set_scan(self, scan):
        self._scan = scan

    def scan_engine(self):
        return "f32" if self._scan == 1 else "i8"

    def set_ef(self, ef):
        self.ef = ef

    def set_search_width(self, width):
        self.width = width

    def stats(self):
        return dict(self._st)

    def stats_reset(self):
        self._st = self._zero()

    def drop(self):
        FakeSpace.spaces.pop(self.name, None)

    def __len__(self):
        return self.X.shape[0]


class FakeSearcher:
    def __init__(self, row0, B, k, device, space=None, stream=None, exchange=None):
        self.space, self.k = space, k
        self.ids = torch.empty((B, k), dtype=torch.int64)
        self.dst = torch.empty((B, k), dtype=torch.float32)
        self.cnt = torch.empty((B,), dtype=torch.int32)

    def knn(self, q):
        self.space.knn_device(q, self.k, self.ids, self.dst, self.cnt)
        return self.ids, self.dst, self.cnt


class FakeLib:
    @staticmethod
    def ehx_init(dev, n):
        return 0

    @staticmethod
    def ehx_gen_rows_device(stream, seed, row0, n, d, normalize, ptr):
        rows = pyoracle.gen_rows(seed, row0, n, d, normalize=bool(normalize))
        C.memmove(ptr.value, rows.ctypes.data, rows.nbytes)
        return 0

    @staticmethod
    def ehx_gen_manifold_rows_device(stream, seed, row0, n, d, latent, normalize, ptr):
        rows = pyoracle.gen_manifold_rows(seed, row0, n, d, latent, normalize=bool(normalize))
        C.memmove(ptr.value, rows.ctypes.data, rows.nbytes)
        return 0




def _np_view(ptr, dtype, count):
    return np.frombuffer((C.c_char * (count * np.dtype(dtype).itemsize)).from_address(ptr), dtype=dtype)


def _merge_strided(stream, nq, k, n_lists, ids, ids_stride, dst, dst_stride, cnt, cnt_stride, o_ids, o_dst, o_cnt):
    """ehx_merge_topk_strided_device on host memory: k-way merge by (distance, id) of n_lists lists per query"""
    val = lambda p: p.value if hasattr(p, "value") else int(p)  # noqa: E731
    oi, od, oc = _np_view(val(o_ids), np.int64, nq * k), _np_view(val(o_dst), np.float32, nq * k), _np_view(val(o_cnt), np.int32, nq)
    for q in range(nq):
        items = []
        for g in range(n_lists):
            c = int(_np_view(val(cnt) + g * cnt_stride, np.int32, nq)[q])
            gi = _np_view(val(ids) + g * ids_stride, np.int64, nq * k)[q * k:q * k + c]
            gd = _np_view(val(dst) + g * dst_stride, np.float32, nq * k)[q * k:q * k + c]
            items += list(zip(gd.tolist(), gi.tolist()))
        items.sort()
        c = min(k, len(items))
        oc[q] = c
        for j in range(c):
            od[q * k + j], oi[q * k + j] = items[j]
    return 0


FakeLib.ehx_merge_topk_strided_device = staticmethod(_merge_strided)


class _Direct:
    """monkeypatch look-alike for a process that keeps the stand-ins for its whole life"""
    @staticmethod
    def setitem(d, k, v):
        d[k] = v

    @staticmethod
    def setattr(obj, name, v):
        setattr(obj, name, v)


def install(mp=None, real_sharded=False):
    """Put the stand-ins in place of embeddinghub_amd / torch.cuda.  real_sharded: keep the product's own
    embeddinghub_amd/sharded.py (row partition, packed all-gather, merge call) on top of the stand-in library —
    what the two-rank launcher test runs."""
    mp = mp or _Direct
    ehx = types.ModuleType("embeddinghub_amd")
    for name, v in dict(METRIC_L2SQ=0, METRIC_IP=1, METRIC_COSINE=2, SCAN_AUTO=0, SCAN_F32=1, SCAN_F16=2, DTYPE_F32=0,
                        DTYPE_F16=1, MODE_FLAT=0, MODE_GRAPH=1, SEED_CORPUS=20250211, SEED_QUERY=20250212).items():
        setattr(ehx, name, v)
    ehx.Space = FakeSpace
    lib = types.ModuleType("embeddinghub_amd._lib")
    lib.load = lambda: FakeLib
    lib.check = lambda rc: None
    ehx._lib = lib
    mp.setitem(sys.modules, "embeddinghub_amd", ehx)
    mp.setitem(sys.modules, "embeddinghub_amd._lib", lib)
    if real_sharded:
        import importlib.util
        root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        spec = importlib.util.spec_from_file_location("embeddinghub_amd.sharded",
                                                      os.path.join(root, "embeddinghub_amd", "sharded.py"))
        sharded = importlib.util.module_from_spec(spec)
        mp.setitem(sys.modules, "embeddinghub_amd.sharded", sharded)
        spec.loader.exec_module(sharded)
    else:
        sharded = types.ModuleType("embeddinghub_amd.sharded")
        sharded.shard_range = lambda rows, G, rank: (0, rows)
        sharded.ShardedSearcher = FakeSearcher
        mp.setitem(sys.modules, "embeddinghub_amd.sharded", sharded)
    n_dev = int(os.environ.get("EHX_STANDIN_DEVICES", "1"))
    mp.setattr(torch.cuda, "is_available", lambda: True)
    mp.setattr(torch.cuda, "set_device", lambda d: None)
    mp.setattr(torch.cuda, "synchronize", lambda *a: None)
    mp.setattr(torch.cuda, "mem_get_info", lambda *a: (200 << 30, 288 << 30))
    mp.setattr(torch.cuda, "device_count", lambda: n_dev)
    mp.setattr(torch.cuda, "current_stream",
               lambda *a: types.SimpleNamespace(cuda_stream=0, synchronize=lambda: None, wait_event=lambda e: None))
    fake_stream = lambda *a, **kw: types.SimpleNamespace(wait_event=lambda e: None, cuda_stream=0,  # noqa: E731
                                                         synchronize=lambda: None)
    fake_event = lambda *a, **kw: types.SimpleNamespace(record=lambda *s: None)  # noqa: E731
    import contextlib
    mp.setattr(torch.cuda, "Stream", fake_stream)
    mp.setattr(torch.cuda, "Event", fake_event)
    mp.setattr(torch.cuda, "stream", lambda s: contextlib.nullcontext())
    no_dev = lambda f: (lambda *a, **kw: f(*a, **{k: v for k, v in kw.items() if k != "device"}))  # noqa: E731
    mp.setattr(torch, "empty", no_dev(torch.empty))
    mp.setattr(torch, "zeros", no_dev(torch.zeros))
    mp.setattr(torch, "tensor", no_dev(torch.tensor))
    return ehx
