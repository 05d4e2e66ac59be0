"""GPU parity tests of graph-mode search (k_graph.hip) against the oracle's HNSW on the SAME graph:
the oracle builds the graph (reference defaults M=16, efC=200, seed=100, sequential insertion), the
engine imports it (ehx_graph_import) and must return bit-identical ids and distances for every ef,
with identical work counters (n_dist = rows fetched, n_hops = nodes expanded)."""
import numpy as np
import pytest

from oracle import pyoracle

pytestmark = pytest.mark.gpu

ehx = pytest.importorskip("embeddinghub_amd")

METRICS = [(ehx.METRIC_L2SQ, pyoracle.METRIC_L2), (ehx.METRIC_IP, pyoracle.METRIC_IP),
           (ehx.METRIC_COSINE, pyoracle.METRIC_COSINE)]


def _build(n, d, em, om, seed=0, M=16):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n, d)).astype(np.float32)
    h = pyoracle.Hnsw(d, om, n, M=M)
    h.add_rows(X)
    s = ehx.Space.unique("graph", d, metric=em, mode=ehx.MODE_GRAPH, M=M, initial_capacity=n)
    s.set_batch(["k%d" % i for i in range(n)], X)
    l0, lv, upper = h.export_graph()
    s.graph_import(l0, lv, upper, h.enterpoint, h.maxlevel)
    return X, h, s, rng


@pytest.mark.parametrize("em,om", METRICS)
@pytest.mark.parametrize("n,d,nq,k,efs", [
    (3000, 64, 40, 10, (10, 50, 200)),      # several levels, reference default ef first
    (10000, 128, 64, 10, (10, 200)),        # BASELINE config 1 shape (10k x 128), ef = 10 and 200
    (1500, 768, 16, 10, (10, 64)),          # BASELINE dims
    (700, 20, 9, 5, (10,)),                 # SIMD4 distance path
    (500, 19, 7, 3, (16,)),                 # residual distance path
])
def test_strict_parity_on_oracle_graph(n, d, nq, k, efs, em, om):
    X, h, s, rng = _build(n, d, em, om, seed=n + d)
    Q = rng.standard_normal((nq, d)).astype(np.float32)
    for ef in efs:
        h.set_ef(ef)
        s.set_ef(ef)
        s.stats_reset()
        labels, dists, counts, _, st = h.search_batch(Q, k, threads=1)
        ids, dist, cnt = s.knn(Q, k)
        np.testing.assert_array_equal(cnt, counts)
        np.testing.assert_array_equal(ids, labels)
        assert dist.tobytes() == dists.tobytes()
        g = s.stats()
        # hnswlib re-evaluates the level-0 entry point at the top of searchBaseLayerST; the engine
        # reuses the descent's final distance, i.e. exactly one row fetch fewer per query
        assert g["n_dist"] == st["n_dist"] - nq, (ef, g["n_dist"], st["n_dist"])
        assert g["n_hops"] == st["n_hops0"] + st["n_hops_up"]
        assert g["bytes_algorithmic"] == ((st["n_dist"] - nq) * d * 4 + st["n_hops0"] * (4 + 4 * 32)
                                          + st["n_hops_up"] * (4 + 4 * 16))
    s.drop()


def test_graph_mode_requires_a_graph_and_flat_agrees_at_high_ef():
    X, h, s, rng = _build(2000, 32, ehx.METRIC_L2SQ, pyoracle.METRIC_L2, seed=5)
    Q = rng.standard_normal((32, 32)).astype(np.float32)
    s.set_ef(400)
    ids, dist, cnt = s.knn(Q, 10)
    truth, tdist, _ = pyoracle.exhaustive(X, Q, 10, pyoracle.METRIC_L2)
    recall = np.mean([len(set(ids[i]) & set(truth[i])) / 10 for i in range(32)])
    assert recall >= 0.99
    # every returned distance is the canonical distance of the returned id
    for i in range(4):
        for j in range(10):
            assert dist[i, j] == np.float32(pyoracle.dist(pyoracle.METRIC_L2, Q[i], X[ids[i, j]]))
    s.set("new-key", X[0])  # the graph no longer covers every row
    with pytest.raises(ehx.EhxError) as e:
        s.knn(Q, 10)
    assert e.value.code == ehx._lib.EUNSUPPORTED
    s.drop()


def test_small_graphs_and_k_larger_than_index():
    for n in (1, 2, 5, 40):
        X, h, s, rng = _build(n, 8, ehx.METRIC_L2SQ, pyoracle.METRIC_L2, seed=n)
        Q = rng.standard_normal((3, 8)).astype(np.float32)
        labels, dists, counts, _, _ = h.search_batch(Q, 10, threads=1)
        ids, dist, cnt = s.knn(Q, 10)
        np.testing.assert_array_equal(cnt, counts)
        for i in range(3):
            c = int(cnt[i])
            np.testing.assert_array_equal(ids[i, :c], labels[i, :c])
            assert dist[i, :c].tobytes() == dists[i, :c].tobytes()
        s.drop()
