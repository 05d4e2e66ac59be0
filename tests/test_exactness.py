"""GPU tests of the exactness contract of the flat path: an EHX_OK result is the certified exhaustive top-k —
never a silently approximate one.

 * the certification margin is a worst-case bound in d (cert_margin, ehx_kernels.h): large d, all-positive
   (ReLU-like) rows and near-ties must still come out byte-identical to the oracle;
 * any number of queries the matrix-core scans cannot certify is answered by the exhaustive canonical pass
   (it used to give up above 32 per call);
 * the fp16 scan copy stays current while EHX_SCAN_F32 is selected (stale copy after set_scan);
 * a dropped space leaves a tombstone: late users of the handle get EHX_ENOTFOUND, not freed memory.
"""
import threading

import numpy as np
import pytest

from oracle import pyoracle

pytestmark = pytest.mark.gpu

ehx = pytest.importorskip("embeddinghub_amd")

METRICS = [(ehx.METRIC_L2SQ, pyoracle.METRIC_L2), (ehx.METRIC_IP, pyoracle.METRIC_IP),
           (ehx.METRIC_COSINE, pyoracle.METRIC_COSINE)]


def _keys(n, p="k"):
    return ["%s%d" % (p, i) for i in range(n)]


def _check(space, X, Q, k, ometric):
    ids, dist, cnt = space.knn(Q, k)
    oids, odist, ocnt = pyoracle.exhaustive(X, Q, k, ometric)
    np.testing.assert_array_equal(cnt, ocnt)
    for i in range(Q.shape[0]):
        c = int(cnt[i])
        assert list(ids[i, :c]) == list(oids[i, :c]), "query %d ids differ" % i
        assert dist[i, :c].tobytes() == odist[i, :c].tobytes(), "query %d distances not bit-exact" % i
    assert space.stats()["n_uncertified"] == 0


@pytest.mark.parametrize("n,d,nq", [(3000, 4096, 24), (1200, 16384, 12)])
@pytest.mark.parametrize("em,om", METRICS)
@pytest.mark.parametrize("scan", [ehx.SCAN_AUTO, ehx.SCAN_F32])
def test_high_d_all_positive_rows_and_near_ties(n, d, nq, em, om, scan):
    """same-sign products (the worst case of fp32 chain accumulation: error ~ d * 2^-24 * sum|q x|) at d = 4096
    and 16384, with clusters of near-duplicates whose distances differ by a few ulps"""
    rng = np.random.default_rng(d + n)
    X = np.abs(rng.standard_normal((n, d))).astype(np.float32)
    base = X[:40].copy()
    for j in range(4):  # 4 x 40 near-copies: ties finer than any matrix-core arithmetic resolves
        X[100 + 40 * j:140 + 40 * j] = base * np.float32(1.0 + 3e-7 * (j + 1))
    X[300:310] = base[:10]  # exact duplicates (ids break the ties)
    Q = np.abs(rng.standard_normal((nq, d))).astype(np.float32)
    Q[:8] = base[:8] * np.float32(1.0 + 1e-7)
    s = ehx.Space.unique("hid", d, metric=em, scan=scan)
    s.set_batch(_keys(n), X)
    _check(s, X, Q, 10, om)
    s.drop()


@pytest.mark.parametrize("em,om", METRICS)
def test_many_uncertifiable_queries_are_all_answered_exactly(em, om):
    """A duplicate cluster wider than any candidate list hit by more than 32 queries of one call: no matrix-core
    stage can certify them (the k'-th candidate ties the k-th); every one must come back exact through the
    exhaustive canonical pass — none may be returned uncertified."""
    rng = np.random.default_rng(5)
    n, d, k = 4000, 96, 10
    X = rng.standard_normal((n, d)).astype(np.float32)
    X[500:700] = X[500]      # 200 identical rows
    X[900:1100] = X[900]
    Q = rng.standard_normal((100, d)).astype(np.float32)
    Q[:45] = X[500]
    Q[45:80] = X[900] * np.float32(1.5)
    s = ehx.Space.unique("dup", d, metric=em)
    s.set_batch(_keys(n), X)
    _check(s, X, Q, k, om)
    st = s.stats()
    assert st["n_exhaustive"] >= 80, st
    assert st["n_uncertified"] == 0
    s.drop()


def test_scan_copy_stays_current_while_f32_is_selected():
    rng = np.random.default_rng(11)
    d, k = 200, 10
    s = ehx.Space.unique("stale", d, metric=ehx.METRIC_COSINE, initial_capacity=256)
    X = rng.standard_normal((300, d)).astype(np.float32)
    s.set_batch(_keys(300), X)
    s.set_scan(ehx.SCAN_F32)
    X2 = rng.standard_normal((5000, d)).astype(np.float32)   # several capacity doublings + new rows
    s.set_batch(_keys(5000, "n"), X2)
    upd = rng.standard_normal((50, d)).astype(np.float32)    # and rows updated in place
    s.set_batch(_keys(50), upd)
    Xall = np.concatenate([X, X2])
    Xall[:50] = upd
    Q = np.concatenate([X2[4000:4016] + np.float32(1e-3), upd[:8], rng.standard_normal((8, d)).astype(np.float32)])
    _check(s, Xall, Q, k, pyoracle.METRIC_COSINE)
    s.set_scan(ehx.SCAN_AUTO)
    s.stats_reset()
    _check(s, Xall, Q, k, pyoracle.METRIC_COSINE)
    assert s.stats()["n_filter_queries"] == Q.shape[0], "the filter engine did not run after switching back"
    s.drop()


def test_dropped_space_is_a_tombstone():
    L = ehx._lib.load()
    s = ehx.Space.unique("tomb", 8)
    s.set("a", np.arange(8, dtype=np.float32))
    h = s._h
    name = s.name
    s.drop()
    # the stale handle fails cleanly on every kind of entry point
    stale = ehx.Space(name, 8, _handle=h)
    for call in (lambda: stale.knn(np.zeros((1, 8), np.float32), 1), lambda: stale.set("b", np.zeros(8)),
                 lambda: stale.get("a"), lambda: len(stale), lambda: stale.stats(), lambda: stale.freeze()):
        with pytest.raises(ehx.EhxError) as e:
            call()
        assert e.value.code == ehx._lib.ENOTFOUND
    # the name is free again
    s2 = ehx.Space(name, 8)
    s2.set("a", np.ones(8, dtype=np.float32))
    assert s2.knn_keys(np.ones(8, np.float32), 1) == [["a"]]
    s2.drop()
    assert L is not None


def test_drop_races_with_searches_and_writes():
    """DeleteSpace while NearestNeighbor / Set handlers still hold the handle (rpc/server.py runs them
    concurrently): every call either completes or fails with ENOTFOUND; nothing crashes or hangs."""
    rng = np.random.default_rng(3)
    d = 64
    for rep in range(4):
        s = ehx.Space.unique("race", d, metric=ehx.METRIC_COSINE)
        s.set_batch(_keys(2000), rng.standard_normal((2000, d)).astype(np.float32))
        errs, stop = [], threading.Event()

        def searcher(t):
            q = np.random.default_rng(100 + t).standard_normal((1, d)).astype(np.float32)
            while not stop.is_set():
                try:
                    s.knn(q, 5)
                except ehx.EhxError as e:
                    if e.code != ehx._lib.ENOTFOUND:
                        errs.append(e)
                    return

        def writer(t):
            i, wr = 0, np.random.default_rng(200 + t)
            while not stop.is_set():
                try:
                    s.set("w%d-%d" % (t, i), wr.standard_normal(d).astype(np.float32))
                    i += 1
                except ehx.EhxError as e:
                    if e.code != ehx._lib.ENOTFOUND:
                        errs.append(e)
                    return

        ths = ([threading.Thread(target=searcher, args=(t,)) for t in range(6)] +
               [threading.Thread(target=writer, args=(t,)) for t in range(2)])
        for t in ths:
            t.start()
        h = s._h
        ehx._lib.check(ehx._lib.load().ehx_space_drop(h))
        stop.set()
        for t in ths:
            t.join(timeout=60)
            assert not t.is_alive(), "a thread hung across the drop"
        assert not errs, errs
