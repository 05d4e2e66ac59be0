"""GPU parity tests of the flat (exhaustive MFMA) kNN path, through the C ABI, against the oracle.

Bar (north_star): integer ids bit-exact, distances within 1e-4 relative.  The engine's canonical
re-rank recomputes distances in the oracle's summation order, so these tests assert BIT-EXACT
distances as well and only fall back to the 1e-4 bound in the message.
"""
import json
import os

import numpy as np
import pytest

from oracle import offline_oracle, pyoracle

pytestmark = pytest.mark.gpu

ehx = pytest.importorskip("embeddinghub_amd")
from embeddinghub_amd import offlinehub  # noqa: E402

METRICS = [(ehx.METRIC_L2SQ, pyoracle.METRIC_L2), (ehx.METRIC_IP, pyoracle.METRIC_IP),
           (ehx.METRIC_COSINE, pyoracle.METRIC_COSINE)]


def _keys(n):
    return ["k%d" % i for i in range(n)]


def _check(space, X, Q, k, ometric):
    ids, dist, cnt = space.knn(Q, k)
    oids, odist, ocnt = pyoracle.exhaustive(X, Q, k, ometric)
    np.testing.assert_array_equal(cnt, ocnt)
    for i in range(Q.shape[0]):
        c = int(cnt[i])
        assert list(ids[i, :c]) == list(oids[i, :c]), "query %d ids differ" % i
        np.testing.assert_allclose(dist[i, :c], odist[i, :c], rtol=1e-4, atol=0)
        assert dist[i, :c].tobytes() == odist[i, :c].tobytes(), "query %d distances not bit-exact" % i
    return ids, dist, cnt


# ---- reference known-answer tests through the engine --------------------------------------------
def _abc():
    s = ehx.Space.unique("abc", 3)
    s.set("a", [0, 1, 0])
    s.set("b", [1, 1, 0])
    s.set("c", [1, 0, 0])
    return s


def test_simple_ann():  # index_test.cc:17-26
    assert _abc().knn_keys([0, 1, 0], 1) == [["a"]]


def test_multi_ann():  # index_test.cc:28-37
    assert _abc().knn_keys([0, 1, 0], 2) == [["a", "b"]]


def test_update_ann():  # index_test.cc:39-49
    s = _abc()
    s.set("a", [0, -1, 0])
    assert s.knn_keys([0, 1, 0], 1) == [["b"]]
    np.testing.assert_array_equal(s.get("a"), np.array([0, -1, 0], dtype=np.float32))


def test_ann_0_items():  # index_test.cc:51-60
    assert _abc().knn_keys([0, 1, 0], 0) == [[]]


def test_rpc_semantics_match_oracle():  # server.cc:172-210
    s = _abc()
    o = pyoracle.AnnIndex(3)
    for k, v in [("a", [0, 1, 0]), ("b", [1, 1, 0]), ("c", [1, 0, 0])]:
        o.set(k, v)
    cases = [dict(num=1, key="a", embedding=[0, 1, 0]), dict(num=1), dict(num=2, key="a"),
             dict(num=2, embedding=[0, 1, 0]), dict(num=1, key="zzz"), dict(num=1, key="c")]
    for c in cases:
        assert ehx.nearest_neighbor_rpc(s, **c) == o.nearest_neighbor_rpc(**c), c


@pytest.mark.parametrize("mode", ["flat", "graph"])
def test_by_key_with_keys_in_one_call_is_the_rpc(mode):
    """ehx_knn_by_key_keys = the NearestNeighbor RPC by key (server.cc:172-210) in one call: the same neighbours as
    ehx_knn_by_key + ehx_key_of per id, their keys in the arena (long keys: the arena grows on EHX_ERANGE), fewer than k
    results on a small space, EHX_ENOTFOUND for an unknown key"""
    rng = np.random.default_rng(5)
    n, d = 700, 24
    X = rng.standard_normal((n, d)).astype(np.float32)
    keys = ["key-%05d-" % i + "x" * (i % 7 == 0) * 300 for i in range(n)]
    kw = dict(mode=ehx.MODE_GRAPH, build_batch=1) if mode == "graph" else {}
    s = ehx.Space.unique("bykey-" + mode, d, metric=ehx.METRIC_L2SQ, **kw)
    s.set_batch(keys, X)
    if mode == "graph":
        s.set_ef(64)
    for q in (0, 7, 349, 699):
        ids, _ = s.knn_by_key(keys[q], 20)
        assert s.knn_by_key_keys(keys[q], 20) == [s.key_of(i) for i in ids] and len(ids) == 20 and q not in ids
    if mode == "flat":   # (exact: the neighbours of a stored row, itself excluded, are the oracle's)
        oids, _, _ = pyoracle.exhaustive(X, X[349:350], 21, pyoracle.METRIC_L2)
        assert s.knn_by_key_keys(keys[349], 20) == [keys[i] for i in oids[0] if i != 349][:20]
    tiny = ehx.Space.unique("bykey-tiny-" + mode, d, metric=ehx.METRIC_L2SQ, **kw)
    tiny.set_batch(keys[:3], X[:3])
    assert sorted(tiny.knn_by_key_keys(keys[1], 20)) == sorted([keys[0], keys[2]])
    with pytest.raises(ehx.EhxError) as e:
        s.knn_by_key_keys("no such key", 5)
    assert e.value.code == ehx._lib.ENOTFOUND
    s.drop()
    tiny.drop()


def test_get_set_roundtrip_and_errors():  # version_test.cc:16-23, storage_test.cc:17-23
    s = ehx.Space.unique("rt", 5, metric=ehx.METRIC_COSINE)
    v = np.array([1.5, -2.25, 3.0, 0.125, 7.0], dtype=np.float32)
    s.set("x", v)
    np.testing.assert_array_equal(s.get("x"), v)  # cosine spaces keep the RAW vector for Get
    with pytest.raises(ehx.EhxError) as e:
        s.get("missing")
    assert e.value.code == ehx._lib.ENOTFOUND
    s.freeze()
    with pytest.raises(ehx.EhxError) as e:  # server.cc:125-127
        s.set("y", v)
    assert e.value.code == ehx._lib.EIMMUTABLE and "immutable" in str(e.value)
    with pytest.raises(ehx.EhxError) as e:
        ehx.Space(s.name, 5)
    assert e.value.code == ehx._lib.EEXISTS
    assert len(ehx.Space.open(s.name)) == 1
    s.drop()
    with pytest.raises(ehx.EhxError):
        ehx.Space.open(s.name)


# ---- offlinehub_test.py against the drop-in Index ------------------------------------------------
INIT = [("a", [1, 0]), ("b", [0, 1]), ("c", [-1, -1]), ("d", [1, 1])]


def test_offline_index_suite():
    index = offlinehub.Index([], 3)
    index.set("a", [1, 2, 3])
    assert index.get("a") == [1, 2, 3]
    index = offlinehub.Index(INIT, 2)
    assert index.get("a") == [1, 0]
    assert index.nearest_neighbor(2, key="a") == ["d", "b"]  # offlinehub_test.py:63-65
    assert index.nearest_neighbor(2, key="a") == offline_oracle.Index(INIT, 2).nearest_neighbor(2, key="a")
    index.set("a", [5, 5])
    assert index.get("a") == [5, 5]
    index.multiset({"a": [3, 3], "b": [4, 4]})
    assert index.multiget(["a", "b", "c"]) == [[3, 3], [4, 4], [-1, -1]]


def test_offline_capacity_growth_and_total_ties():  # offlinehub_test.py:68-86
    index = offlinehub.Index([], 2)
    for key in list(range(1025)) * 2:
        index.set(str(key), [1, 1])
    assert index.size() == 1025
    embs = [(key, [1, 1]) for key in list(range(1028)) * 2]
    index = offlinehub.Index(embs, 2)
    assert index.size() == 1028
    index = offlinehub.Index([], 2)
    for i in range(0, len(embs), 4):
        index.multiset(embs[i:i + 4])
    assert index.size() == 1028
    # all 1028 distances tie at 0: exhaustive order is (dist, id) -> the first-set keys
    assert index.nearest_neighbor(5, embedding=[1, 1]) == [0, 1, 2, 3, 4]


# ---- Go boundary golden -----------------------------------------------------------------------------
def test_go_vectorstore_golden(golden_dir):  # vectorstore_test.go:121-166
    g = json.load(open(os.path.join(golden_dir, "go_vectorstore.json")))
    X = np.array(g["vectors"], dtype=np.float32)
    q = np.array(g["query"], dtype=np.float32)
    s = ehx.Space.unique("go", 768, metric=ehx.METRIC_COSINE)
    for ent, v in zip(g["entities"], X):
        s.set(ent, v)
        np.testing.assert_array_equal(s.get(ent), v)  # testGetSet: reflect.DeepEqual round trip
    assert s.knn_keys(q, 2) == [["Investor's Business Daily", "Seeking Alpha"]]
    _, dist, _ = s.knn(q, 5)
    np.testing.assert_allclose(np.sort(dist[0]), np.sort(g["float64_cosine_distance"]), rtol=1e-5)
    _check(s, X, q[None, :], 2, pyoracle.METRIC_COSINE)


# ---- randomized parity vs the exhaustive oracle -------------------------------------------------------
@pytest.mark.parametrize("n,d,nq,k", [
    (1000, 128, 37, 10),     # ragged everything
    (4096, 768, 256, 10),    # exact tile multiples
    (5000, 768, 300, 10),    # BASELINE dims, two query tiles
    (777, 3, 5, 4),          # scalar distance path (d <= 4)
    (900, 7, 9, 3),          # SIMD4 residual path
    (600, 19, 9, 3),         # SIMD16 residual path
    (1200, 20, 11, 7),       # SIMD4 path (d % 4 == 0, d % 16 != 0)
    (3000, 1536, 16, 10),    # config-5 dims
    (300, 64, 1, 48),        # max k
    (70000, 32, 64, 10),     # many tiles per chunk, every CU busy
])
@pytest.mark.parametrize("em,om", METRICS)
@pytest.mark.parametrize("scan", [ehx.SCAN_AUTO, ehx.SCAN_F32])  # fp16 filter + fp32 re-rank | fp32 scan only
def test_random_parity(n, d, nq, k, em, om, scan):
    rng = np.random.default_rng(n * 31 + d)
    X = rng.standard_normal((n, d)).astype(np.float32)
    Q = rng.standard_normal((nq, d)).astype(np.float32)
    s = ehx.Space.unique("rand", d, metric=em, scan=scan)
    s.set_batch(_keys(n), X)
    assert len(s) == n
    _check(s, X, Q, k, om)
    st = s.stats()
    assert st["n_uncertified"] == 0 and st["n_rows"] == n
    if scan == ehx.SCAN_AUTO and nq == 1:   # one query on a small shard: straight to the exhaustive canonical pass
        assert st["n_exhaustive"] == 1 and st["n_i8_queries"] == 0 and st["n_filter_queries"] == 0
    elif scan == ehx.SCAN_AUTO:   # a filter engine answered first: int8 (>= 16 Ki rows, d <= 2048), else fp16
        assert (st["n_i8_queries"] or st["n_filter_queries"]) == nq
    else:
        assert st["n_filter_queries"] == 0 and st["n_i8_queries"] == 0
    assert st["n_filter_fallback"] <= nq // 4 and st["n_i8_fallback"] <= nq // 4  # re-runs must stay the exception
    s.drop()


@pytest.mark.parametrize("em,om", METRICS)
@pytest.mark.parametrize("n,d,k", [(10000, 128, 10), (5000, 768, 10), (300, 19, 48), (40000, 96, 100), (3, 8, 10)])
def test_single_query_on_a_small_shard_takes_the_exhaustive_pass(n, d, k, em, om):
    """The reference's own request shape — ONE query per NearestNeighbor call (server.cc:172-210; BASELINE configs[0]) —
    on a shard small enough to be read in less time than the batch engines need to start: answered by the exhaustive
    canonical pass alone, exactly; two or more queries in a call take the engines as before."""
    rng = np.random.default_rng(n + d + k)
    X = rng.standard_normal((n, d)).astype(np.float32)
    Q = rng.standard_normal((3, d)).astype(np.float32)
    Q[0] = X[n // 2] + np.float32(1e-3) * rng.standard_normal(d).astype(np.float32)
    s = ehx.Space.unique("single", d, metric=em)
    s.set_batch(_keys(n), X)
    for i in range(3):
        s.stats_reset()
        _check(s, X, Q[i:i + 1], k, om)
        st = s.stats()
        assert st["n_exhaustive"] == 1 and st["n_i8_queries"] == 0 and st["n_filter_queries"] == 0 and st["n_uncertified"] == 0
    s.stats_reset()
    _check(s, X, Q, k, om)          # three queries in one call: the engine chain (k <= 48) or the paged pass (k > 48)
    st = s.stats()
    assert (st["n_i8_queries"] or st["n_filter_queries"]) == 3 if k <= 48 else st["n_exhaustive"] == 3
    s.drop()


@pytest.mark.parametrize("em,om", METRICS)
def test_filter_scale_invariance_and_fallback(em, om):
    """The fp16 filter normalises rows and queries, so wildly scaled data must give the oracle's answers;
    near-duplicate rows defeat its certification and must be answered by the fp32 re-run."""
    rng = np.random.default_rng(77)
    n, d = 4000, 96
    X = rng.standard_normal((n, d)).astype(np.float32)
    X *= np.exp(rng.uniform(-12, 12, size=(n, 1))).astype(np.float32)  # row norms over ten decades
    Q = (rng.standard_normal((24, d)) * np.exp(rng.uniform(-8, 8, size=(24, 1)))).astype(np.float32)
    s = ehx.Space.unique("scale", d, metric=em)
    s.set_batch(_keys(n), X)
    _check(s, X, Q, 10, om)
    s.drop()
    # 40 rows within 5 % of each other around the query: more near-ties than the filter keeps candidates,
    # its lower bounds cannot separate them — such queries must fall back to the fp32 scan and stay exact
    base = rng.standard_normal(d).astype(np.float32)
    Y = rng.standard_normal((3000, d)).astype(np.float32)
    Y[:40] = base * (1.0 + 5e-2 * rng.standard_normal((40, d)).astype(np.float32))
    s = ehx.Space.unique("dups", d, metric=em)
    s.set_batch(_keys(3000), Y)
    Qd = np.stack([base, Y[5], rng.standard_normal(d).astype(np.float32)])
    _check(s, Y, Qd, 10, om)
    st = s.stats()
    assert st["n_uncertified"] == 0
    if em != ehx.METRIC_IP:  # (raw inner products of these rows are spread far enough for the filter)
        assert st["n_filter_fallback"] >= 1
    s.drop()
    # 200 rows within 1e-4 of each other: their distances to the query differ by less than the fp32 rounding
    # of ANY matrix-core arithmetic, so not even the fp32 scan can certify — the exhaustive canonical pass
    # has to produce the oracle's answer
    Z = rng.standard_normal((3000, d)).astype(np.float32)
    Z[:200] = base * (1.0 + 1e-4 * rng.standard_normal((200, d)).astype(np.float32))
    s = ehx.Space.unique("neardups", d, metric=em)
    s.set_batch(_keys(3000), Z)
    _check(s, Z, np.stack([base, Z[5]]), 10, om)
    st = s.stats()
    assert st["n_uncertified"] == 0
    if em != ehx.METRIC_IP:
        assert st["n_exhaustive"] >= 1
    s.drop()


def test_incremental_set_update_and_growth_match_oracle():
    rng = np.random.default_rng(5)
    d = 48
    X = rng.standard_normal((700, d)).astype(np.float32)
    s = ehx.Space.unique("inc", d)  # starts at capacity 128 (index.h:21), doubles (index.cc:29-32)
    for i in range(300):
        s.set("k%d" % i, X[i])
    s.set_batch(_keys(700)[300:], X[300:])
    # update-in-place keeps the label (index.cc:21-35)
    X[17] = rng.standard_normal(d).astype(np.float32)
    s.set("k17", X[17])
    X[650] = X[3]  # exact duplicate row -> distance tie resolved by id
    s.set("k650", X[650])
    Q = np.concatenate([X[:4], rng.standard_normal((8, d)).astype(np.float32)])
    _check(s, X, Q, 10, pyoracle.METRIC_L2)
    ids, dist = s.knn_by_key("k3", 5)
    assert 3 not in ids and ids[0] == 650 and dist[0] == 0.0


def test_fewer_rows_than_k_and_empty_space():
    s = ehx.Space.unique("small", 8, metric=ehx.METRIC_IP)
    Q = np.ones((3, 8), dtype=np.float32)
    ids, dist, cnt = s.knn(Q, 5)
    assert list(cnt) == [0, 0, 0]
    X = np.arange(24, dtype=np.float32).reshape(3, 8)
    s.set_batch(_keys(3), X)
    _check(s, X, Q, 5, pyoracle.METRIC_IP)  # count 3 < k
    _check(s, X, Q, 49, pyoracle.METRIC_IP)  # k beyond one scan pass: the paged exhaustive pass, count 3
    with pytest.raises(ehx.EhxError) as e:
        s.knn(Q, 1025)
    assert e.value.code == ehx._lib.EUNSUPPORTED


@pytest.mark.parametrize("em,om", METRICS)
@pytest.mark.parametrize("k", [49, 64, 65, 200])
def test_large_k_is_served_by_the_paged_exhaustive_pass(em, om, k):
    """EHX_MAX_K < k <= EHX_MAX_K_PAGED (the reference accepts any num, index.cc:39-52): exact, in pages of 64."""
    rng = np.random.default_rng(k)
    n, d = 20000, 40
    X = rng.standard_normal((n, d)).astype(np.float32)
    X[100:140] = X[7]  # ties across a page boundary are ordered by id
    Q = np.concatenate([X[7:8], rng.standard_normal((4, d)).astype(np.float32)])
    s = ehx.Space.unique("bigk", d, metric=em)
    s.set_batch(_keys(n), X)
    _check(s, X, Q, k, om)
    assert s.stats()["n_exhaustive"] == Q.shape[0]
    s.drop()


def test_nan_and_inf_rows_do_not_poison_results():
    rng = np.random.default_rng(2)
    X = rng.standard_normal((500, 16)).astype(np.float32)
    X[5, 3] = np.nan
    X[9, 1] = np.inf
    s = ehx.Space.unique("nan", 16)
    s.set_batch(_keys(500), X)
    Q = rng.standard_normal((4, 16)).astype(np.float32)
    ids, dist, cnt = s.knn(Q, 10)
    assert (cnt == 10).all() and np.isfinite(dist).all()
    assert 5 not in ids and 9 not in ids


# ---- fp16 row storage (BASELINE.json configs[5]) -------------------------------------------------
# An F16 space must behave exactly like an F32 space that was fed the rows rounded to binary16
# (round-to-nearest-even), so the oracle is run on the rounded rows and everything stays bit-exact.
@pytest.mark.parametrize("n,d,nq,k", [
    (1000, 32, 8, 10),
    (5000, 100, 33, 10),     # dims not a multiple of the 32-element stage
    (3000, 1536, 16, 10),    # config-5 dims
    (70000, 64, 64, 10),     # many tiles per chunk
])
@pytest.mark.parametrize("em,om", METRICS)
@pytest.mark.parametrize("scan", [ehx.SCAN_AUTO, ehx.SCAN_F32])
def test_fp16_rows_parity(n, d, nq, k, em, om, scan):
    rng = np.random.default_rng(n * 17 + d)
    X = rng.standard_normal((n, d)).astype(np.float32)
    Q = rng.standard_normal((nq, d)).astype(np.float32)
    Xh = X.astype(np.float16).astype(np.float32)
    s = ehx.Space.unique("h16", d, metric=em, dtype=ehx.DTYPE_F16, scan=scan)
    s.set_batch(_keys(n), X)
    _check(s, Xh, Q, k, om)
    assert s.get("k7").tobytes() == Xh[7].tobytes()  # Get returns the stored (rounded) row
    # update in place + growth
    X2 = rng.standard_normal((300, d)).astype(np.float32)
    s.set_batch(_keys(n + 200)[n - 100:], X2)
    Xh = np.concatenate([Xh[:n - 100], X2.astype(np.float16).astype(np.float32)])
    _check(s, Xh, Q, k, om)
    st = s.stats()
    assert st["n_uncertified"] == 0 and st["n_rows"] == n + 200
    s.drop()


def test_fp16_synthetic_fill():
    n, d = 20000, 768
    s = ehx.Space.unique("h16s", d, metric=ehx.METRIC_COSINE, dtype=ehx.DTYPE_F16, initial_capacity=n)
    s.fill_synthetic(ehx.SEED_CORPUS, 0, n, True)
    Xh = pyoracle.gen_rows(ehx.SEED_CORPUS, 0, n, d, normalize=True).astype(np.float16).astype(np.float32)
    Q = pyoracle.gen_rows(ehx.SEED_QUERY, 0, 32, d, normalize=True)
    _check(s, Xh, Q, 10, pyoracle.METRIC_COSINE)
    s.drop()   # (fp16 rows behind a graph: tests/test_graph_parity.py)


# ---- BASELINE-size properties (size-independent checks at the bench workload's shape) -------------
def test_full_size_split_invariance_and_sample_vs_oracle():
    """1M x 768 cosine, batch 1024 (BASELINE config 2): (a) kNN over the whole index == merge of the
    kNN over two half indexes (ids are row ids of the same EHX-GAUSS-1 rows); (b) results sorted by
    (dist, id); (c) a sample of queries against the oracle's exhaustive scan of oracle-generated rows."""
    n, d, nq, k = 1_000_000, 768, 1024, 10
    whole = ehx.Space.unique("full", d, metric=ehx.METRIC_COSINE, initial_capacity=n)
    whole.fill_synthetic(ehx.SEED_CORPUS, 0, n, True)
    lo = ehx.Space.unique("lo", d, metric=ehx.METRIC_COSINE, initial_capacity=n // 2)
    lo.fill_synthetic(ehx.SEED_CORPUS, 0, n // 2, True)
    hi = ehx.Space.unique("hi", d, metric=ehx.METRIC_COSINE, initial_capacity=n // 2)
    hi.fill_synthetic(ehx.SEED_CORPUS, n // 2, n // 2, True)
    Q = pyoracle.gen_rows(ehx.SEED_QUERY, 0, nq, d, normalize=True)
    ids, dist, cnt = whole.knn(Q, k)
    ids_lo, dist_lo, _ = lo.knn(Q, k)
    ids_hi, dist_hi, _ = hi.knn(Q, k)
    assert (cnt == k).all() and whole.stats()["n_uncertified"] == 0
    for i in range(nq):
        cand = sorted(zip(dist_lo[i].tolist() + dist_hi[i].tolist(),
                          ids_lo[i].tolist() + (ids_hi[i] + n // 2).tolist()))[:k]
        assert [c[1] for c in cand] == ids[i].tolist(), i
        assert np.array([c[0] for c in cand], dtype=np.float32).tobytes() == dist[i].tobytes()
        assert all((dist[i, j], ids[i, j]) <= (dist[i, j + 1], ids[i, j + 1]) for j in range(k - 1))
    # oracle on a sample: 8 queries x the first 200k rows (regenerated on the host, bit-identical)
    m = 200_000
    Xs = pyoracle.gen_rows(ehx.SEED_CORPUS, 0, m, d, normalize=True)
    sub = ehx.Space.unique("sub", d, metric=ehx.METRIC_COSINE, initial_capacity=m)
    sub.fill_synthetic(ehx.SEED_CORPUS, 0, m, True)
    sids, sdist, _ = sub.knn(Q[:8], k)
    oids, odist, _ = pyoracle.exhaustive(Xs, Q[:8], k, pyoracle.METRIC_COSINE)
    np.testing.assert_array_equal(sids, oids)
    assert sdist.tobytes() == odist.tobytes()
    for s in (whole, lo, hi, sub):
        s.drop()


# ---- one query per call: the reference's request shape (server.cc:172-210; BASELINE configs[0]) ------------------
# A single query from host memory against a small flat shard takes ONE launch (k_flat.hip: single_query_kernel: the
# query is read from host-visible memory, the last workgroup writes the answer back and raises a flag).  Everything the
# three-launch path guaranteed must hold: the oracle's ids and distance bytes, counts when rows < k, NaN rows skipped,
# rows visible as soon as their Set returned, in-place updates seen by the next call.
@pytest.mark.parametrize("n,d,k", [(1, 3, 10), (5, 100, 10), (64, 128, 10), (1000, 33, 48), (10000, 128, 10),
                                   (20000, 768, 10), (3000, 1536, 5), (70000, 64, 10)])
@pytest.mark.parametrize("em,om", METRICS)
@pytest.mark.parametrize("dtype", [ehx.DTYPE_F32, ehx.DTYPE_F16])
def test_one_query_per_call_is_the_oracle(n, d, k, em, om, dtype):
    rng = np.random.default_rng(n * 31 + d)
    X = rng.standard_normal((n, d)).astype(np.float32)
    Q = rng.standard_normal((12, d)).astype(np.float32)
    Q[0] = X[n // 2]                                   # a stored row itself
    s = ehx.Space.unique("one", d, metric=em, dtype=dtype)
    s.set_batch(_keys(n), X)
    Xs = X.astype(np.float16).astype(np.float32) if dtype == ehx.DTYPE_F16 else X
    s.stats_reset()
    for i in range(Q.shape[0]):
        _check(s, Xs, Q[i:i + 1], k, om)
    st = s.stats()
    assert st["n_queries"] == Q.shape[0] and st["n_exhaustive"] == Q.shape[0] and st["n_uncertified"] == 0
    # a batch in between (other engines, other buffers), then single queries again
    _check(s, Xs, Q, k, om)
    _check(s, Xs, Q[3:4], k, om)
    # an in-place update and an append are seen by the very next call
    X2 = Xs.copy()
    X2[0] = np.float32(0.5) * Q[5]
    s.set("k0", X2[0])
    new = (Q[6] * np.float32(1.0001)).astype(np.float32)
    s.set("fresh", new)
    if dtype == ehx.DTYPE_F16:
        X2[0] = X2[0].astype(np.float16).astype(np.float32)
        new = new.astype(np.float16).astype(np.float32)
    X2 = np.concatenate([X2, new[None, :]])
    for i in (5, 6, 7):
        _check(s, X2, Q[i:i + 1], k, om)
    s.drop()


def test_one_query_per_call_skips_nan_rows_and_handles_zero_queries():
    rng = np.random.default_rng(5)
    X = rng.standard_normal((300, 16)).astype(np.float32)
    X[7, 2] = np.nan
    X[11, 0] = np.inf
    s = ehx.Space.unique("one-nan", 16, metric=ehx.METRIC_COSINE)
    s.set_batch(_keys(300), X)
    q = rng.standard_normal((1, 16)).astype(np.float32)
    ids, dist, cnt = s.knn(q, 10)
    assert cnt[0] == 10 and np.isfinite(dist).all() and 7 not in ids and 11 not in ids
    ids, dist, cnt = s.knn(np.zeros((1, 16), np.float32), 10)       # a zero query: cosine distance 1 to everything
    assert cnt[0] == 10
    s.drop()
