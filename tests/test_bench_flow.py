"""Dry run of bench.py's main() on the CPU (no GPU, no engine): the engine, torch.cuda and the sharded searcher are
replaced by small numpy / oracle stand-ins so that the WHOLE control flow runs — timed region, exactness block (ten
batches against the 'fp32 engine', the oracle check), CPU baseline, the optional legs under their time budget and
watchdog, the one JSON line.  What this pins is bench.py itself (names, order, the JSON contract of the driver), not
the engine: the stand-in answers every kNN with the oracle's exhaustive scan."""
import ctypes as C
import importlib.util
import io
import json
import os
import sys
import types
from contextlib import redirect_stdout

import numpy as np
import pytest
import torch

from oracle import pyoracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class FakeSpace:
    spaces = {}

    def __init__(self, name, dims, metric=0, mode=0, initial_capacity=0, shards=0, dtype=0, build_batch=0, **kw):
        self.name, self.dims, self.metric, self.mode = name, dims, metric, mode
        self.X = np.zeros((0, dims), dtype=np.float32)
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
        self.X = np.concatenate([self.X, pyoracle.gen_rows(seed, row0, n, self.dims, normalize=bool(normalize))])

    def set_batch(self, keys, X):
        self.X = np.concatenate([self.X, np.asarray(X, dtype=np.float32)])

    def knn(self, Q, k):
        Q = np.asarray(Q, dtype=np.float32).reshape(-1, self.dims)
        ids, dist, cnt = pyoracle.exhaustive(self.X, Q, k, self._om())
        self._st["scan_launches"] += 1
        self._st["n_queries"] += Q.shape[0]
        self._st["n_dist"] += 100 * Q.shape[0]
        self._st["n_hops"] += 5 * Q.shape[0]
        self._st["bytes_algorithmic"] += 100 * Q.shape[0] * self.dims * 4
        self._st["n_i8_queries"] += Q.shape[0]
        return ids, dist, cnt

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

    def stats(self):
        return dict(self._st)

    def stats_reset(self):
        self._st = self._zero()

    def drop(self):
        FakeSpace.spaces.pop(self.name, None)

    def __len__(self):
        return self.X.shape[0]


class FakeSearcher:
    def __init__(self, row0, B, k, device, space=None, stream=None):
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


@pytest.fixture
def bench_on_stand_ins(monkeypatch):
    ehx = types.ModuleType("embeddinghub_amd")
    for name, v in dict(METRIC_L2SQ=0, METRIC_IP=1, METRIC_COSINE=2, SCAN_AUTO=0, SCAN_F32=1, SCAN_F16=2, DTYPE_F32=0,
                        DTYPE_F16=1, MODE_FLAT=0, MODE_GRAPH=1, SEED_CORPUS=20250211, SEED_QUERY=20250212).items():
        setattr(ehx, name, v)
    ehx.Space = FakeSpace
    lib = types.ModuleType("embeddinghub_amd._lib")
    lib.load = lambda: FakeLib
    lib.check = lambda rc: None
    ehx._lib = lib
    sharded = types.ModuleType("embeddinghub_amd.sharded")
    sharded.shard_range = lambda rows, G, rank: (0, rows)
    sharded.ShardedSearcher = FakeSearcher
    monkeypatch.setitem(sys.modules, "embeddinghub_amd", ehx)
    monkeypatch.setitem(sys.modules, "embeddinghub_amd._lib", lib)
    monkeypatch.setitem(sys.modules, "embeddinghub_amd.sharded", sharded)
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "set_device", lambda d: None)
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a: None)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    monkeypatch.setattr(torch.cuda, "current_stream",
                        lambda *a: types.SimpleNamespace(cuda_stream=0, synchronize=lambda: None))
    real_empty = torch.empty
    monkeypatch.setattr(torch, "empty", lambda *a, **kw: real_empty(*a, **{k: v for k, v in kw.items() if k != "device"}))
    spec = importlib.util.spec_from_file_location("bench_flow_module", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for var in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(var, raising=False)

    def run(argv):
        monkeypatch.setattr(sys, "argv", ["bench.py"] + argv)
        buf = io.StringIO()
        with redirect_stdout(buf):
            bench.main()
        lines = [l for l in buf.getvalue().splitlines() if l.strip()]
        assert len(lines) == 1, "bench.py must print exactly ONE line on stdout: %r" % lines
        return json.loads(lines[0])
    run.module = bench
    return run


SMALL = ["--rows", "3000", "--dims", "32", "--batch", "16", "--steps", "3", "--warmup", "1", "--check-queries", "8",
         "--cpu-sample-rows", "3000", "--cpu-sample-queries", "8", "--cpu-hnsw-rows", "500", "--graph-batches", "2",
         "--graph-efs", "10,20"]


def test_default_flow_prints_one_json_line_with_the_contract_fields(bench_on_stand_ins):
    r = bench_on_stand_ins(SMALL + ["--graph-rows", "1000", "--structured-rows", "600"])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in r, key
    assert r["n_gpus"] == 1 and r["steps"] == 3 and r["warmup"] == 1 and r["higher_is_better"] is True
    assert r["vs_baseline"] is None and r["data"] == "synthetic" and "workload" in r["config"]
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(r["roofline"])
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(r["cpu_baseline"]) and r["cpu_baseline"]["kind"] == "port"
    ex = r["exactness"]
    assert ex["filter_vs_f32_engine_identical"] is True and ex["filter_vs_f32_engine_queries"] == 4 * 16  # all 4 resident batches
    assert ex["recall_at_10"] == 1.0 and ex["ids_identical_to_oracle"] and ex["dist_bytes_identical_to_oracle"]
    assert ex["oracle_rows"] == 3000 and "oracle_partial" not in ex
    assert r["graph_path"]["recall_vs_ef"][0]["ef"] == 10 and "operating_point" in r["graph_path"]
    assert "graph_path_structured" in r and r["optional_legs_skipped"] == {}
    assert "host_pointer_path" in r and "f32_scan_engine" in r


def test_a_spent_time_budget_skips_the_optional_legs_and_says_so(bench_on_stand_ins):
    r = bench_on_stand_ins(SMALL + ["--graph-rows", "1000", "--structured-rows", "600", "--time-budget", "0"])
    assert "graph_path" not in r and "graph_path_structured" not in r
    assert set(r["optional_legs_skipped"]) == {"graph_path", "graph_path_structured"}
    assert "cpu_baseline" in r and r["exactness"]["recall_at_10"] == 1.0   # the required legs ran


def test_a_slow_host_gets_the_prefix_check(bench_on_stand_ins, monkeypatch):
    # an oracle that can only afford its first chunk: the run must still finish, flagged as partial, still consistent
    bench = bench_on_stand_ins.module
    real = bench.oracle_truth
    monkeypatch.setattr(bench, "oracle_truth", lambda *a, **kw: real(*a, **dict(kw, max_seconds=0.0)))
    r = bench_on_stand_ins(SMALL + ["--rows", "1200000", "--dims", "8", "--graph-rows", "0", "--structured-rows", "0",
                                    "--no-cpu-baseline", "--check-queries", "4"])
    ex = r["exactness"]
    assert ex.get("oracle_partial") is True and 0 < ex["oracle_rows"] < 1200000
    assert ex["ids_identical_to_oracle"] is True and ex["recall_at_10"] == 1.0


def test_a_stuck_optional_leg_is_abandoned_by_the_watchdog(bench_on_stand_ins, monkeypatch):
    import threading
    import time
    bench = bench_on_stand_ins.module
    exits = []

    class QuickTimer(threading.Timer):          # the watchdog's interval shrunk to 0.2 s for the test
        def __init__(self, interval, function, *a, **kw):
            super().__init__(0.2, function, *a, **kw)
    monkeypatch.setattr(bench.threading, "Timer", QuickTimer)
    monkeypatch.setattr(bench.os, "_exit", lambda code: exits.append(code))   # (the real one ends the process)
    real_fill = FakeSpace.fill_synthetic

    def slow_fill(self, seed, row0, n, normalize):
        if self.mode == 1:                      # the graph leg's index: "stuck"
            time.sleep(0.8)
        real_fill(self, seed, row0, n, normalize)
    monkeypatch.setattr(FakeSpace, "fill_synthetic", slow_fill)
    r = bench_on_stand_ins(SMALL + ["--graph-rows", "1000", "--structured-rows", "0", "--no-cpu-baseline"])
    assert exits == [0]
    assert "graph_path" not in r and "watchdog" in r["optional_legs_skipped"]["graph_path"]
    assert r["exactness"]["recall_at_10"] == 1.0 and r["value"] > 0       # the headline part is all there


def test_multi_rank_flow_rank0_prints_the_line(bench_on_stand_ins, monkeypatch):
    """the path torchrun drives (WORLD_SIZE > 1): barriers, MAX-reduced time, no single-GPU legs — collectives stubbed"""
    import torch.distributed as dist
    calls = []
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("LOCAL_RANK", "0")
    monkeypatch.setattr(dist, "init_process_group", lambda *a, **kw: calls.append("init"))
    monkeypatch.setattr(dist, "barrier", lambda *a, **kw: calls.append("barrier"))
    monkeypatch.setattr(dist, "all_reduce", lambda t, op=None: calls.append("all_reduce"))
    monkeypatch.setattr(dist, "destroy_process_group", lambda *a, **kw: calls.append("destroy"))
    real_tensor, real_device = torch.tensor, torch.device
    monkeypatch.setattr(torch, "tensor", lambda *a, **kw: real_tensor(*a, **{k: v for k, v in kw.items() if k != "device"}))
    monkeypatch.setattr(torch, "device", lambda *a, **kw: real_device("cpu"))
    # the stand-in shard holds every row (shard_range stub), so the merged answer is the exact one
    r = bench_on_stand_ins(SMALL + ["--gpus", "2"])
    assert r["n_gpus"] == 2 and "row-shard x2" in r["config"]["parallelism"]
    assert "graph_path" not in r and "cpu_baseline" not in r          # N = 1 legs
    assert r["exactness"]["recall_at_10"] == 1.0
    assert calls[0] == "init" and calls[-1] == "destroy" and "all_reduce" in calls and calls.count("barrier") >= 3
