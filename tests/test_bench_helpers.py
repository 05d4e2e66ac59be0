"""CPU tests of bench.py's host-side helpers (no GPU): the oracle-truth reducer with its time box, the prefix
consistency check used when the host cannot scan every row in time, the CPU-baseline sample sizing, the single JSON
emit.  (The oracle is the checker here, as in bench.py itself.)"""
import importlib.util
import io
import json
import os
import types
from contextlib import redirect_stdout

import numpy as np

from oracle import pyoracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(bench)


def _args(**kw):
    base = dict(rows=40_000, dims=32, k=10, time_budget=300.0, cpu_sample_rows=1_000_000, cpu_sample_queries=16,
                cpu_hnsw_rows=0)
    base.update(kw)
    return types.SimpleNamespace(**base)


def test_oracle_truth_is_the_exhaustive_scan_and_can_be_time_boxed():
    args = _args()
    Q = pyoracle.gen_rows(20250212, 0, 6, args.dims, normalize=True)
    ids, dist, covered = bench.oracle_truth(args.rows, args.dims, Q, 10, pyoracle.METRIC_COSINE, chunk=7_000)  # ragged last chunk
    assert covered == args.rows
    X = pyoracle.gen_rows(20250211, 0, args.rows, args.dims, normalize=True)
    oids, odist, _ = pyoracle.exhaustive(X, Q, 10, pyoracle.METRIC_COSINE)
    np.testing.assert_array_equal(ids, oids.astype(np.uint64))
    assert dist.tobytes() == odist.tobytes()
    # a spent time box stops after the first chunk and says how far it got
    _, _, part = bench.oracle_truth(args.rows, args.dims, Q, 10, pyoracle.METRIC_COSINE, chunk=7_000, max_seconds=0.0)
    assert part == 7_000


def test_oracle_truth_of_an_fp16_space_scans_the_rounded_rows_and_the_check_fails_closed():
    """an fp16-row space stores the generated rows rounded to binary16 — the oracle must scan those;
    and ANY id / distance-byte mismatch ends the run, whatever the recall"""
    import pytest
    rows, dims = 30_000, 64
    Q = pyoracle.gen_rows(20250212, 0, 6, dims, normalize=True)
    Xh = pyoracle.gen_rows(20250211, 0, rows, dims, normalize=True).astype(np.float16).astype(np.float32)
    oids, odist, _ = pyoracle.exhaustive(Xh, Q, 10, pyoracle.METRIC_COSINE)
    ids, dist, covered = bench.oracle_truth(rows, dims, Q, 10, pyoracle.METRIC_COSINE, f16_rows=True, chunk=8_000)
    assert covered == rows and dist.tobytes() == odist.tobytes()
    np.testing.assert_array_equal(ids, oids.astype(np.uint64))
    res = bench.oracle_check(rows, dims, 10, pyoracle.METRIC_COSINE, True, Q, oids, odist, None)
    assert res["ids_identical_to_oracle"] and res["dist_bytes_identical_to_oracle"] and res["recall_at_10"] == 1.0
    # the answer of an engine that holds the UNROUNDED rows: same ids nearly everywhere, other bytes -> the run ends
    X = pyoracle.gen_rows(20250211, 0, rows, dims, normalize=True)
    wids, wdist, _ = pyoracle.exhaustive(X, Q, 10, pyoracle.METRIC_COSINE)
    with pytest.raises(SystemExit) as e:
        bench.oracle_check(rows, dims, 10, pyoracle.METRIC_COSINE, True, Q, wids, wdist, None)
    assert "EXACTNESS FAILURE" in str(e.value)
    # ... in prefix mode too (a host that only gets through the first chunk)
    real = bench.oracle_truth
    bench.oracle_truth = lambda *a, **kw: real(*a, **dict(kw, chunk=8_000, max_seconds=0.0))
    try:
        ok = bench.oracle_check(rows, dims, 10, pyoracle.METRIC_COSINE, True, Q, oids, odist, 0.0)
        assert ok["oracle_partial"] and ok["ids_identical_to_oracle"]
        with pytest.raises(SystemExit):
            bench.oracle_check(rows, dims, 10, pyoracle.METRIC_COSINE, True, Q, wids, wdist, 0.0)
    finally:
        bench.oracle_truth = real


def test_prefix_consistency_accepts_the_exact_answer_and_flags_wrong_ones():
    args = _args()
    Q = pyoracle.gen_rows(20250212, 0, 8, args.dims, normalize=True)
    full_i, full_d, _ = bench.oracle_truth(args.rows, args.dims, Q, 10, pyoracle.METRIC_COSINE, chunk=10_000)
    half_i, half_d, cov = bench.oracle_truth(20_000, args.dims, Q, 10, pyoracle.METRIC_COSINE, chunk=10_000)
    ok, frac = bench.prefix_consistent(full_i, full_d, half_i, half_d, cov)
    assert ok and frac == 1.0
    # an engine that lost a prefix row which belongs to the global top-k
    q, j = next((q, j) for q in range(8) for j in range(9) if full_i[q, j] < cov)
    bad_i, bad_d = full_i.copy(), full_d.copy()
    bad_i[q, j] = args.rows - 1 - q
    ok, frac = bench.prefix_consistent(bad_i, bad_d, half_i, half_d, cov)
    assert not ok and frac < 1.0
    # an engine whose distance for a prefix row differs in the last bit
    bad_i, bad_d = full_i.copy(), full_d.copy()
    bad_d[q, j] = np.nextafter(bad_d[q, j], np.float32(0))
    ok, _ = bench.prefix_consistent(bad_i, bad_d, half_i, half_d, cov)
    assert not ok


def test_cpu_baseline_sample_is_sized_and_reported():
    out = bench.cpu_baseline(_args(rows=120_000, dims=16), pyoracle.METRIC_COSINE)
    assert out["sample_rows"] == 120_000 and out["cores"] == bench.host_cores() >= 1
    assert out["kind"] == "port" and out["value"] > 0 and "hnsw" not in out
    assert abs(out["value"] - out["value_on_sample"]) < 0.02 * out["value"]   # the sample IS the whole index here


def test_the_json_line_is_emitted_once():
    bench._EMIT = __import__("threading").Lock()
    buf = io.StringIO()
    with redirect_stdout(buf):
        bench.emit({"metric": "m", "value": 1})
        bench.emit({"metric": "m", "value": 2})     # a watchdog firing after the main thread printed: nothing more
    lines = buf.getvalue().splitlines()
    assert len(lines) == 1 and json.loads(lines[0])["value"] == 1 and len(lines[0]) <= bench.LINE_LIMIT


def test_engine_agreement_helpers_over_ten_batches():
    import torch
    assert bench.check_batches(2, 12, 10) == list(range(2, 12))
    assert bench.check_batches(5, 25, 10) == list(range(5, 15))
    assert bench.check_batches(1, 3, 10) == [1, 2, 0]                 # fewer resident batches than wanted: each once
    g = torch.Generator().manual_seed(0)
    data = {b: (torch.randint(0, 1000, (8, 10), generator=g), torch.rand((8, 10), generator=g),
                torch.full((8,), 10, dtype=torch.int32)) for b in range(12)}
    buf = [None]

    def step(b):  # one reused output buffer, as the searcher has
        buf[0] = tuple(t.clone() for t in data[b])
        return buf[0]
    kept = bench.keep_results(step, bench.check_batches(2, 12, 10), lambda: None)
    assert sorted(kept) == list(range(2, 12))
    assert bench.first_disagreement(step, kept, torch, lambda: None) is None
    ids, dst, cnt = data[7]
    dst2 = dst.clone()
    dst2[3, 4] = torch.nextafter(dst2[3, 4], torch.tensor(2.0))       # one distance off by one ulp
    data[7] = (ids, dst2, cnt)
    assert bench.first_disagreement(step, kept, torch, lambda: None) == 7


def test_host_callers_are_long_lived_threads_that_split_the_batches_and_report_every_call():
    """bench.HostCallers: `callers` threads started once, thread t serves batches t, t + callers, ... of every run() in
    order; call_ms / call_summary describe the LAST run; an exception in a caller thread is re-raised by run(); close()
    ends the threads."""
    import threading

    class Space:
        def __init__(self):
            self.calls = []          # (thread name, batch marker)
            self.fail_on = None
            self.lock = threading.Lock()

        def knn_into(self, Q, k, ids, dist, cnt):
            b = int(Q[0, 0])
            if self.fail_on == b:
                raise RuntimeError("boom %d" % b)
            with self.lock:
                self.calls.append((threading.current_thread().name, b))
            ids[:] = b

    hq = [np.full((4, 8), b, np.float32) for b in range(6)]
    sp = Space()
    hc = bench.HostCallers(sp, hq, 3, 2)
    try:
        names0 = {t.name for t in hc._threads}
        assert len(names0) == 2
        el = hc.run([0, 1, 2, 3, 4])
        assert el > 0
        by_thread = {}
        for name, b in sp.calls:
            by_thread.setdefault(name, []).append(b)
        assert sorted(by_thread.values()) == [[0, 2, 4], [1, 3]]          # t, t + 2, ... in order, one thread each
        assert set(by_thread) == names0                                   # ... the threads that were started once
        assert [s_["calls"] for s_ in hc.call_summary()] == [3, 2]
        assert int(hc.out[0][0][0, 0]) == 4 and int(hc.out[1][0][0, 0]) == 3   # every thread's own output arrays
        sp.calls.clear()
        hc.run([5])                                                       # fewer batches than threads: one sits out
        assert [b for _, b in sp.calls] == [5] and {n for n, _ in sp.calls} <= names0
        assert [s_["calls"] for s_ in hc.call_summary()] == [1]           # (of the last run only)
        sp.fail_on = 1
        try:
            hc.run([0, 1, 2])
        except RuntimeError as e:
            assert "boom 1" in str(e)
        else:
            raise AssertionError("the caller thread's exception was swallowed")
        sp.fail_on = None
        hc.run([2, 3])                                                    # still usable afterwards
    finally:
        started = list(hc._threads)
        hc.close()
    assert len(started) == 2 and not any(t.is_alive() for t in started) and hc._threads == []
    one = bench.HostCallers(sp, hq, 3, 1)                                 # one caller: the calling thread itself
    sp.calls.clear()
    one.run([0, 1])
    assert {n for n, _ in sp.calls} == {threading.current_thread().name} and one._threads == []


def test_cpu_quota_helpers():
    """round 5: the cgroup throttle counters bench.py records beside its timed regions, and the BLAS pool cap"""
    st = bench.cpu_throttle_stat()
    assert st is None or (len(st) == 2 and st[0] >= 0 and st[1] >= 0)
    assert bench.throttle_delta(None) is None
    if st is not None:
        d = bench.throttle_delta(st)
        assert set(d) == {"times", "ms"} and d["times"] >= 0 and d["ms"] >= 0.0
    n = bench.limit_blas_threads()
    assert n is None or 1 <= n <= max(1, bench.host_cores())
