"""Thread-safety contract of the boundary: the reference serialises everything behind one mutex
(server.cc:175) and the Go materialisation runner calls Table.Set from 500 goroutines per chunk
(runner/copy.go:34, 146-161); the engine must stay correct under concurrent Set / Nearest and
coalesces concurrent single-query searches into device batches."""
import threading

import numpy as np
import pytest

from oracle import pyoracle

pytestmark = pytest.mark.gpu
ehx = pytest.importorskip("embeddinghub_amd")


def test_concurrent_single_query_searches_are_coalesced_and_exact():
    rng = np.random.default_rng(0)
    n, d, k, T, per = 20000, 64, 10, 32, 8
    X = rng.standard_normal((n, d)).astype(np.float32)
    s = ehx.Space.unique("conc", d)
    s.set_batch(["k%d" % i for i in range(n)], X)
    Q = rng.standard_normal((T * per, d)).astype(np.float32)
    oids, odist, _ = pyoracle.exhaustive(X, Q, k, pyoracle.METRIC_L2)
    out = [None] * (T * per)
    errs = []

    def worker(t):
        try:
            for j in range(per):
                i = t * per + j
                out[i] = s.knn(Q[i], k)
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    ths = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    assert not errs, errs
    for i in range(T * per):
        ids, dist, cnt = out[i]
        assert cnt[0] == k
        np.testing.assert_array_equal(ids[0], oids[i])
        assert dist[0].tobytes() == odist[i].tobytes()
    s.drop()


def test_concurrent_sets_then_search():
    rng = np.random.default_rng(1)
    n, d, T = 4000, 32, 16
    X = rng.standard_normal((n, d)).astype(np.float32)
    s = ehx.Space.unique("conc-set", d, metric=ehx.METRIC_COSINE)
    errs = []

    def writer(t):
        try:
            for i in range(t, n, T):
                s.set("e%d" % i, X[i])
                if i % 7 == 0:  # a reader interleaved with the writers
                    s.knn(X[i], 3)
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    ths = [threading.Thread(target=writer, args=(t,)) for t in range(T)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    assert not errs, errs
    assert len(s) == n
    # ids were assigned in arrival order: map them back through the keys
    order = np.array([int(s.key_of(i)[1:]) for i in range(n)])
    assert sorted(order) == list(range(n))
    Q = rng.standard_normal((16, d)).astype(np.float32)
    ids, dist, _ = s.knn(Q, 5)
    oids, odist, _ = pyoracle.exhaustive(X[order], Q, 5, pyoracle.METRIC_COSINE)
    np.testing.assert_array_equal(ids, oids)
    assert dist.tobytes() == odist.tobytes()
    for i in (0, 1, n - 1):
        np.testing.assert_array_equal(s.get("e%d" % i), X[i])
    s.drop()
