"""GPU tests of row sharding BEHIND THE C ABI (ehx_params.shards): one process, G shards — on a
single-GPU box they all live on device 0, which exercises the routing (row g -> shard g % G, local row g / G), the
concurrent per-shard searches, the gather and the merge exactly as on an 8-GPU node.  A sharded space must be
indistinguishable from an unsharded one: same ids (global, dense, first-Set order), same distance bytes."""
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


@pytest.mark.parametrize("G", [2, 3, 8])
@pytest.mark.parametrize("em,om", METRICS)
def test_sharded_space_equals_the_oracle(G, em, om):
    rng = np.random.default_rng(G)
    n, d, nq, k = 9001, 96, 70, 10
    X = rng.standard_normal((n, d)).astype(np.float32)
    Q = rng.standard_normal((nq, d)).astype(np.float32)
    s = ehx.Space.unique("shd", d, metric=em, shards=G)
    for i0 in range(0, n, 2500):                       # several batches: growth inside every shard
        s.set_batch(_keys(n)[i0:i0 + 2500], X[i0:i0 + 2500])
    assert len(s) == n
    _check(s, X, Q, k, om)
    # upserts keep their global row id; Get returns exactly what was Set, whatever shard holds the row
    upd = rng.standard_normal((40, d)).astype(np.float32)
    s.set_batch(_keys(n)[100:140], upd)
    X[100:140] = upd
    assert len(s) == n
    for i in (0, 99, 100, 139, n - 1):
        assert s.get("k%d" % i).tobytes() == X[i].tobytes()
        assert s.key_of(i) == "k%d" % i
    _check(s, X, Q, k, om)
    assert s.knn_keys(X[7], 1) == [["k7"]]
    st = s.stats()
    assert st["n_rows"] == n and st["n_uncertified"] == 0
    s.drop()


def test_sharded_synthetic_fill_is_the_same_corpus():
    """ehx_fill_synthetic on a sharded space: global row g is generator row g, so the result equals the unsharded
    space's and the oracle's over the oracle-generated rows"""
    n, d, nq, k = 50_000, 768, 64, 10
    s = ehx.Space.unique("shs", d, metric=ehx.METRIC_COSINE, shards=2, initial_capacity=n)
    s.fill_synthetic(ehx.SEED_CORPUS, 0, 30_000, True)
    s.fill_synthetic(ehx.SEED_CORPUS, 30_000, n - 30_000, True)      # a second fill continues the numbering
    X = pyoracle.gen_rows(ehx.SEED_CORPUS, 0, n, d, normalize=True)
    Q = pyoracle.gen_rows(ehx.SEED_QUERY, 0, nq, d, normalize=True)
    _check(s, X, Q, k, pyoracle.METRIC_COSINE)
    assert s.scan_engine() == "i8"                                   # 25 000 rows per shard: the int8 engine
    assert s.get_by_id(12345).tobytes() == X[12345].tobytes()
    s.drop()


def test_sharded_device_entry_point_and_graph_mode():
    import torch
    rng = np.random.default_rng(4)
    n, d, nq, k = 6000, 64, 128, 10
    X = rng.standard_normal((n, d)).astype(np.float32)
    Q = rng.standard_normal((nq, d)).astype(np.float32)
    s = ehx.Space.unique("shdv", d, metric=ehx.METRIC_L2SQ, shards=4)
    s.set_batch(_keys(n), X)
    dq = torch.from_numpy(Q).cuda()
    ids = torch.empty((nq, k), dtype=torch.int64, device="cuda")
    dst = torch.empty((nq, k), dtype=torch.float32, device="cuda")
    cnt = torch.empty((nq,), dtype=torch.int32, device="cuda")
    s.knn_device(dq, k, ids, dst, cnt, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    oids, odist, _ = pyoracle.exhaustive(X, Q, k, pyoracle.METRIC_L2)
    np.testing.assert_array_equal(ids.cpu().numpy().astype(np.uint64), oids)
    assert dst.cpu().numpy().tobytes() == odist.tobytes()
    s.drop()
    # graph mode: every shard builds its own HNSW; with ef >= the shard size the search is exhaustive per shard,
    # so the merged answer is the exact one
    g = ehx.Space.unique("shg", d, metric=ehx.METRIC_L2SQ, mode=ehx.MODE_GRAPH, shards=2, ef=800)
    g.set_batch(_keys(1200), X[:1200])
    gi, gd, gc = g.knn(Q[:16], k)
    oi, od, _ = pyoracle.exhaustive(X[:1200], Q[:16], k, pyoracle.METRIC_L2)
    if not np.array_equal(gi, oi):
        # (seen once in ~15 runs of round 3 and never reproduced: say which side is at fault if it happens again)
        again = g.knn(Q[:16], k)[0]
        missing = sorted(set(oi.ravel().tolist()) - set(gi.ravel().tolist()))
        raise AssertionError("sharded graph search differs from the exact answer: ids never returned %s; a second "
                             "search of the same graph %s" % (missing, "agrees with the exact answer (the SEARCH is "
                             "not deterministic)" if np.array_equal(again, oi) else "differs too (the GRAPH misses links)"))
    assert gd.tobytes() == od.tobytes()
    g.drop()


def test_shards_are_hidden_and_dropped_with_their_parent():
    s = ehx.Space.unique("shh", 8, shards=2)
    s.set("a", np.ones(8, dtype=np.float32))
    with pytest.raises(ehx.EhxError) as e:
        ehx.Space.open(s.name + "\x010")
    assert e.value.code == ehx._lib.ENOTFOUND
    name = s.name
    s.drop()
    t = ehx.Space(name, 8, shards=2)      # the name (and the shard names) are free again
    t.set("b", np.ones(8, dtype=np.float32))
    assert t.knn_keys(np.ones(8, np.float32), 1) == [["b"]]
    t.drop()


@pytest.mark.parametrize("k", [65, 100, 300])
def test_sharded_space_serves_large_k_like_an_unsharded_one(k):
    """sharded spaces capped k at 64 while unsharded ones serve 1024 — every shard pages its own exhaustive
    pass beyond 48 results, and the merge walks the G lists (merge_lists_walk_kernel) beyond 64"""
    rng = np.random.default_rng(k)
    n, d, nq = 2600, 40, 9
    X = rng.standard_normal((n, d)).astype(np.float32)
    X[1700] = X[3]                                    # a cross-shard exact tie: (distance, id) order decides
    Q = np.concatenate([X[:3], rng.standard_normal((nq - 3, d)).astype(np.float32)])
    s = ehx.Space.unique("shk", d, metric=ehx.METRIC_L2SQ, shards=3)
    s.set_batch(_keys(n), X)
    _check(s, X, Q, k, pyoracle.METRIC_L2)
    s.drop()
    tiny = ehx.Space.unique("shk2", d, metric=ehx.METRIC_COSINE, shards=4)   # fewer rows than k: count < k
    tiny.set_batch(_keys(50), X[:50])
    _check(tiny, X[:50], Q, k, pyoracle.METRIC_COSINE)
    tiny.drop()


def test_dropping_a_sharded_space_under_concurrent_searches():
    """the parent is marked dropped (under its exclusive lock) BEFORE its shards are released, so a search
    racing with the drop either completes on live shards or reports NOT_FOUND — never reads a released shard"""
    import threading
    rng = np.random.default_rng(12)
    n, d = 4000, 32
    X = rng.standard_normal((n, d)).astype(np.float32)
    Q = rng.standard_normal((16, d)).astype(np.float32)
    oids, _, _ = pyoracle.exhaustive(X, Q, 5, pyoracle.METRIC_L2)
    for rnd in range(3):
        s = ehx.Space.unique("shdrop", d, metric=ehx.METRIC_L2SQ, shards=4)
        s.set_batch(_keys(n), X)
        outcomes, stop = [], threading.Event()

        def searcher():
            while not stop.is_set():
                try:
                    ids, _, _ = s.knn(Q, 5)
                    outcomes.append("ok" if np.array_equal(ids, oids) else "WRONG")
                except ehx.EhxError as e:
                    outcomes.append("gone" if e.code == ehx._lib.ENOTFOUND else "ERR %d" % e.code)
                    return
        ts = [threading.Thread(target=searcher) for _ in range(4)]
        for t in ts:
            t.start()
        import time
        time.sleep(0.05 * (rnd + 1))
        s.drop()
        stop.set()
        for t in ts:
            t.join(timeout=30)
            assert not t.is_alive()
        assert set(outcomes) <= {"ok", "gone"}, set(outcomes)
        assert "gone" in outcomes or all(o == "ok" for o in outcomes)


def test_sharded_space_takes_the_round6_additions():
    """ehx_fill_manifold on a sharded space is the same corpus as on an unsharded one (global row g = generator row g), and
    a sharded GRAPH space walks its shards with the width it was given (ehx_params.search_width at creation,
    ehx_space_set_search_width afterwards, handed down to every shard): with ef >= the shard size either walk is
    exhaustive per shard, so the merged answer must be the exact one for every width."""
    n, d, R, nq, k = 20_000, 96, 12, 32, 10
    s = ehx.Space.unique("shm", d, metric=ehx.METRIC_COSINE, shards=2, initial_capacity=n)
    s.fill_manifold(ehx.SEED_CORPUS, 0, 12_000, R, True)
    s.fill_manifold(ehx.SEED_CORPUS, 12_000, n - 12_000, R, True)
    X = pyoracle.gen_manifold_rows(ehx.SEED_CORPUS, 0, n, d, R, normalize=True)
    Q = pyoracle.gen_manifold_rows(ehx.SEED_QUERY, 0, nq, d, R, normalize=True)
    _check(s, X, Q, k, pyoracle.METRIC_COSINE)
    assert s.get_by_id(n - 1).tobytes() == X[n - 1].tobytes()
    s.drop()
    m = 1600
    g = ehx.Space.unique("shgw", d, metric=ehx.METRIC_COSINE, mode=ehx.MODE_GRAPH, shards=2, ef=1000, search_width=4)
    g.set_batch(_keys(m), X[:m])
    oi, od, _ = pyoracle.exhaustive(X[:m], Q, k, pyoracle.METRIC_COSINE)
    for width in (4, 2, 1):
        g.set_search_width(width)
        gi, gd, gc = g.knn(Q, k)
        np.testing.assert_array_equal(gi, oi, err_msg="width %d" % width)
        assert gd.tobytes() == od.tobytes()
    with pytest.raises(Exception):
        g.set_search_width(3)
    g.drop()
