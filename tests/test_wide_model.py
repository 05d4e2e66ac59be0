"""oracle/wide_walk_model.py (the CPU statement of the engine's wide graph walk, used by tests/test_graph_wide.py) is pinned
to the oracle: with ONE expansion per step it must be the oracle's own HNSW search (oracle/hnsw_oracle.hpp, hnswlib's
searchKnn restated) — ids, distance bytes and rows evaluated — on the oracle's graph; with more it must still return
sorted, distinct, canonical results and terminate within the same ef bound."""
import numpy as np
import pytest

from oracle import pyoracle
from oracle.wide_walk_model import wide_search


@pytest.mark.parametrize("om", [pyoracle.METRIC_L2, pyoracle.METRIC_IP, pyoracle.METRIC_COSINE])
def test_model_with_one_expansion_per_step_is_the_oracles_search(om):
    rng = np.random.default_rng(3)
    n, d, nq, k = 2500, 24, 12, 10
    X = rng.standard_normal((n, d)).astype(np.float32)
    Q = rng.standard_normal((nq, d)).astype(np.float32)
    h = pyoracle.Hnsw(d, om, n)
    h.add_rows(X)
    l0, lv, upper = h.export_graph()
    ids, dist, _ = pyoracle.exhaustive(X, Q, n, om)
    D = np.empty((nq, n), dtype=np.float32)
    np.put_along_axis(D, ids.astype(np.int64), dist, axis=1)
    for ef in (10, 64):
        h.set_ef(ef)
        labels, dists, counts, _, st = h.search_batch(Q, k, threads=1)
        tot = 0
        for i in range(nq):
            m_ids, m_dist, c = wide_search(l0, upper, h.enterpoint, h.maxlevel, D[i], ef, 1, k)
            np.testing.assert_array_equal(m_ids, labels[i])
            assert m_dist.tobytes() == dists[i].tobytes()
            assert c["steps"] == c["n_hops0"]
            tot += c["n_dist"]
        assert tot == st["n_dist"] - nq      # (the oracle evaluates the level-0 entry point twice, the engines once)
        for P in (2, 4):
            for i in range(nq):
                m_ids, m_dist, c = wide_search(l0, upper, h.enterpoint, h.maxlevel, D[i], ef, P, k)
                assert len(set(m_ids.tolist())) == k
                assert all((m_dist[j], m_ids[j]) < (m_dist[j + 1], m_ids[j + 1]) for j in range(k - 1))
                assert all(m_dist[j].tobytes() == D[i, int(m_ids[j])].tobytes() for j in range(k))
                assert c["steps"] <= c["n_hops0"] <= P * c["steps"]
