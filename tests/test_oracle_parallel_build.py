"""The oracle's PARALLEL bulk build (hnswlib's multi-threaded add_items, restated) — used by bench.py's CPU baseline only.
Its graph depends on thread timing, so nothing here compares it bit for bit: the tests hold it to hnswlib's invariants, to
the recall of the sequential build, and make sure the sequential path (THE oracle) is what it was."""
import hashlib

import numpy as np
import pytest

from oracle import pyoracle


def _recall(h, Q, truth, k, ef):
    h.set_ef(ef)
    ids, _, _, _, _ = h.search_batch(Q, k, threads=4)
    return float(np.mean([len(set(ids[i]) & set(truth[i])) / k for i in range(Q.shape[0])]))


@pytest.mark.parametrize("metric", [pyoracle.METRIC_L2, pyoracle.METRIC_COSINE])
def test_parallel_build_keeps_the_invariants_and_the_recall_of_the_sequential_build(metric):
    n, d, nq, k = 6000, 32, 128, 10
    X = pyoracle.gen_rows(11, 0, n, d, normalize=False)
    Q = pyoracle.gen_rows(12, 0, nq, d, normalize=False)
    truth, _, _ = pyoracle.exhaustive(X, Q, k, metric)
    seq = pyoracle.Hnsw(d, metric, n)
    seq.add_rows(X)
    par = pyoracle.Hnsw(d, metric, n)
    par.add_rows(X[:100])                                    # an index that already holds rows ...
    par.add_rows_parallel(X[100:], first_label=100, threads=8)   # ... and a parallel batch on top
    assert len(par) == n
    l0, lv, upper = par.export_graph()
    s0, slv, _ = seq.export_graph()
    assert np.array_equal(lv, slv)                           # levels are drawn in row order, as in the sequential build
    deg = l0[:, 0]
    assert (deg >= 1).all() and (deg <= 32).all()
    for i in range(n):
        row = l0[i, 1:1 + deg[i]]
        assert i not in row and len(set(row.tolist())) == len(row) and (row < n).all()
    for (node, level), ids in upper.items():
        assert len(ids) <= 16 and all(lv[j] >= level for j in ids) and node not in ids
    for ef in (10, 50, 200):
        assert abs(_recall(par, Q, truth, k, ef) - _recall(seq, Q, truth, k, ef)) <= 0.02


def test_parallel_build_rejects_known_labels_and_overflow():
    X = pyoracle.gen_rows(3, 0, 50, 8, normalize=False)
    h = pyoracle.Hnsw(8, pyoracle.METRIC_L2, 60)
    h.add_rows(X[:10])
    with pytest.raises(RuntimeError):
        h.add_rows_parallel(X[:5], first_label=5, threads=2)     # labels 5..9 exist
    with pytest.raises(RuntimeError):
        h.add_rows_parallel(np.tile(X, (2, 1)), first_label=10, threads=2)   # 100 rows into 50 free slots
    h.add_rows_parallel(X[10:], first_label=10, threads=3)
    assert len(h) == 50


def test_the_sequential_build_is_unchanged_by_the_parallel_code():
    """a graph + search digest taken BEFORE the parallel build was added to hnsw_oracle.hpp (round 3): the sequential
    path — the oracle every parity test rests on — still produces exactly that"""
    n, d = 3000, 32
    rng = np.random.default_rng(n + d)
    X = rng.standard_normal((n, d)).astype(np.float32)
    h = pyoracle.Hnsw(d, pyoracle.METRIC_L2, n, M=16)
    h.add_rows(X)
    for i in range(0, n, 37):   # updates in place
        h.add(X[(i * 7) % n] * np.float32(1.01), i)
    l0, lv, up = h.export_graph()
    hh = hashlib.sha256()
    hh.update(l0.tobytes())
    hh.update(lv.tobytes())
    for key in sorted(up):
        hh.update(np.asarray(up[key], dtype=np.uint32).tobytes())
    Q = rng.standard_normal((50, d)).astype(np.float32)
    h.set_ef(64)
    ids, dist, cnt, _, st = h.search_batch(Q, 10, threads=1)
    hh.update(ids.tobytes())
    hh.update(dist.tobytes())
    assert (hh.hexdigest()[:16], h.enterpoint, h.maxlevel, st["n_dist"]) == ("07ade06e1ed23261", 115, 2, 49537)


def test_bulk_build_model_with_rounds_of_one_row_is_the_sequential_build():
    """oracle/hnsw_oracle.hpp addPointsRounds — the CPU model of the engine's bulk build (studies only): with one row per
    round it must BE hnswlib's sequential build, list for list"""
    n, d = 2500, 24
    X = np.random.default_rng(1).standard_normal((n, d)).astype(np.float32)
    for metric in (pyoracle.METRIC_L2, pyoracle.METRIC_COSINE):
        a = pyoracle.Hnsw(d, metric, n)
        a.add_rows(X)
        b = pyoracle.Hnsw(d, metric, n)
        b.add_rows_rounds(X, div=10 ** 9, cap=1, threads=3)
        la, va, ua = a.export_graph()
        lb, vb, ub = b.export_graph()
        assert np.array_equal(la, lb) and np.array_equal(va, vb) and (a.enterpoint, a.maxlevel) == (b.enterpoint, b.maxlevel)
        assert set(ua) == set(ub) and all(np.array_equal(ua[key], ub[key]) for key in ua)


def test_bulk_build_model_rounds_keep_the_invariants_and_cost_little_recall():
    n, d, nq, k = 8000, 32, 256, 10
    X = pyoracle.gen_rows(21, 0, n, d, normalize=False)
    Q = pyoracle.gen_rows(22, 0, nq, d, normalize=False)
    truth, _, _ = pyoracle.exhaustive(X, Q, k, pyoracle.METRIC_L2)
    seq = pyoracle.Hnsw(d, pyoracle.METRIC_L2, n)
    seq.add_rows(X)
    mod = pyoracle.Hnsw(d, pyoracle.METRIC_L2, n)
    mod.add_rows_rounds(X, div=64, cap=4096, threads=4)
    l0, lv, upper = mod.export_graph()
    deg = l0[:, 0]
    assert np.array_equal(lv, seq.export_graph()[1]) and (deg >= 1).all() and (deg <= 32).all()
    for i in range(0, n, 7):
        row = l0[i, 1:1 + deg[i]]
        assert i not in row and len(set(row.tolist())) == len(row)
    for ef in (10, 50, 200):
        assert _recall(mod, Q, truth, k, ef) >= _recall(seq, Q, truth, k, ef) - 0.02
