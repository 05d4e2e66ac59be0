"""Pins the CPU oracle (oracle/) against every known-answer test the reference holds for
the kNN path, and against the reference-derived golden of the Go boundary.  CPU only.

Reference anchors:
  embeddinghub/embeddingstore/test/index_test.cc:17-60       (4 ANNIndex tests)
  embeddinghub/sdk/python/test/offlinehub_test.py:29-86      (Index tests)
  embeddinghub/embeddingstore/server.cc:172-210              (RPC semantics)
  provider/vectorstore_test.go:121-166 + test_files/embeddings.csv (Go boundary)
  SURVEY.md Appendix A.1 (libstdc++ minstd_rand0 level generator known answers)
"""
import json
import os

import numpy as np
import pytest

from oracle import offline_oracle, pyoracle
from oracle.pyoracle import METRIC_COSINE, METRIC_IP, METRIC_L2


# ---- index_test.cc -----------------------------------------------------------------
def _abc():
    idx = pyoracle.AnnIndex(3)
    idx.set("a", [0, 1, 0])
    idx.set("b", [1, 1, 0])
    idx.set("c", [1, 0, 0])
    return idx


def test_simple_ann():  # index_test.cc:17-26
    assert _abc().approx_nearest([0, 1, 0], 1) == ["a"]


def test_multi_ann():  # index_test.cc:28-37
    assert _abc().approx_nearest([0, 1, 0], 2) == ["a", "b"]


def test_update_ann():  # index_test.cc:39-49
    idx = _abc()
    idx.set("a", [0, -1, 0])
    assert idx.approx_nearest([0, 1, 0], 1) == ["b"]


def test_ann_0_items():  # index_test.cc:51-60
    assert _abc().approx_nearest([0, 1, 0], 0) == []


# ---- server.cc:172-210 ---------------------------------------------------------------
def test_rpc_semantics():
    idx = _abc()
    assert idx.nearest_neighbor_rpc(1, key="a", embedding=[0, 1, 0])[0] == 3  # both set
    assert idx.nearest_neighbor_rpc(1)[0] == 3  # neither set
    st, keys = idx.nearest_neighbor_rpc(2, key="a")  # k+1, self removed
    assert st == 0 and keys == ["b", "c"]
    st, keys = idx.nearest_neighbor_rpc(2, embedding=[0, 1, 0])
    assert st == 0 and keys == ["a", "b"]
    assert idx.nearest_neighbor_rpc(1, key="zzz")[0] == 5  # NOT_FOUND replaces the reference's UB


# ---- offlinehub_test.py ----------------------------------------------------------------
INIT = [("a", [1, 0]), ("b", [0, 1]), ("c", [-1, -1]), ("d", [1, 1])]


def test_offline_set_get():  # offlinehub_test.py:29-44
    index = offline_oracle.Index([], 3)
    index.set("a", [1, 2, 3])
    assert index.get("a") == [1, 2, 3]
    index = offline_oracle.Index(INIT, 2)
    assert index.get("a") == [1, 0]
    index.set("a", [5, 5])
    assert index.get("a") == [5, 5]


def test_offline_multiset_multiget():  # offlinehub_test.py:47-60
    index = offline_oracle.Index(INIT, 2)
    index.multiset({"a": [3, 3], "b": [4, 4]})
    assert index.multiget(["a", "b", "c"]) == [[3, 3], [4, 4], [-1, -1]]


def test_offline_nn():  # offlinehub_test.py:63-65
    index = offline_oracle.Index(INIT, 2)
    assert index.nearest_neighbor(2, key="a") == ["d", "b"]


def test_offline_capacity_set():  # offlinehub_test.py:68-72 (total ties, growth past 1024)
    index = offline_oracle.Index([], 2)
    for key in list(range(1025)) * 2:
        index.set(str(key), [1, 1])
    assert index.size() == 1025


def test_offline_capacity_init():  # offlinehub_test.py:75-78
    embs = [(key, [1, 1]) for key in list(range(1028)) * 2]
    index = offline_oracle.Index(embs, 2)
    assert index.size() == 1028


def test_offline_capacity_multiset():  # offlinehub_test.py:81-86
    index = offline_oracle.Index([], 2)
    embs = [(key, [1, 1]) for key in list(range(1028)) * 2]
    for i in range(0, len(embs), 4):
        index.multiset(embs[i:i + 4])
    assert index.size() == 1028


# ---- Go boundary golden -------------------------------------------------------------------
def test_go_vectorstore_golden(golden_dir):
    g = json.load(open(os.path.join(golden_dir, "go_vectorstore.json")))
    X = np.array(g["vectors"], dtype=np.float32)
    q = np.array(g["query"], dtype=np.float32)
    assert X.shape == (5, 768)
    # exhaustive, cosine (redis.go:253 / pinecone.go:251 use COSINE)
    ids, dists, counts = pyoracle.exhaustive(X, q, 2, METRIC_COSINE)
    assert [g["entities"][i] for i in ids[0]] == ["Investor's Business Daily", "Seeking Alpha"]
    np.testing.assert_allclose(dists[0], [0.6887761, 0.7369323], rtol=2e-6)
    # all five distances vs the float64 cross-check
    _, d5, _ = pyoracle.exhaustive(X, q, 5, METRIC_COSINE)
    np.testing.assert_allclose(np.sort(d5[0]), np.sort(g["float64_cosine_distance"]), rtol=1e-5)
    # HNSW path gives the same answer (5 points: graph is complete)
    h = pyoracle.Hnsw(768, METRIC_COSINE, 128)
    h.add_rows(X)
    labels, hd = h.search(q, 2)
    assert [g["entities"][i] for i in labels] == ["Investor's Business Daily", "Seeking Alpha"]
    np.testing.assert_array_equal(hd, dists[0])
    # same order under L2^2 (rows are unit-norm): 1.3775524, 1.4738647
    ids2, d2, _ = pyoracle.exhaustive(X, q, 2, METRIC_L2)
    assert list(ids2[0]) == list(ids[0])
    np.testing.assert_allclose(d2[0], [1.3775524, 1.4738647], rtol=2e-6)


# ---- level generator -----------------------------------------------------
def test_level_generator_known_answers():
    assert list(pyoracle.minstd_first(100, 3)) == [1680700, 330237489, 1203733775]
    lv = pyoracle.levels(100, 16, 32)
    expect = np.zeros(32, dtype=np.int32)
    expect[[11, 29]] = 1
    np.testing.assert_array_equal(lv, expect)
    hist = np.bincount(pyoracle.levels(100, 16, 10**6))
    assert list(hist) == [936998, 59129, 3626, 234, 12, 1]


# ---- distance arithmetic: SSE order, non-fused -------------------------------------------
def _sse_l2(a, b):
    """float32 emulation of L2SqrSIMD16Ext (SSE): 4 strided partial sums, mul then add."""
    acc = np.zeros(4, dtype=np.float32)
    for i in range(0, len(a), 4):
        d = (a[i:i + 4] - b[i:i + 4]).astype(np.float32)
        acc = (acc + (d * d).astype(np.float32)).astype(np.float32)
    return np.float32(np.float32(np.float32(acc[0] + acc[1]) + acc[2]) + acc[3])


def _sse_ip_sum(a, b):
    acc = np.zeros(4, dtype=np.float32)
    for i in range(0, len(a), 4):
        acc = (acc + (a[i:i + 4] * b[i:i + 4]).astype(np.float32)).astype(np.float32)
    return np.float32(np.float32(np.float32(acc[0] + acc[1]) + acc[2]) + acc[3])


def _sse_ip(a, b):
    return np.float32(np.float32(1.0) - _sse_ip_sum(a, b))


@pytest.mark.parametrize("dim", [16, 128, 768, 1536, 4, 12, 20, 3, 2, 7, 19, 35])
def test_distance_bit_exact_vs_numpy_emulation(dim):
    rng = np.random.default_rng(dim)
    for _ in range(5):
        a = rng.standard_normal(dim).astype(np.float32)
        b = rng.standard_normal(dim).astype(np.float32)
        if dim % 4 == 0:
            el2, eip = _sse_l2(a, b), _sse_ip(a, b)
        else:
            # residual variants: SIMD body over the largest multiple of 16 (dim>16) or 4 (dim>4),
            # scalar tail; L2: the two sums added; inner product: see below; dim<=4 scalar
            body = (dim >> 4 << 4) if dim > 16 else ((dim >> 2 << 2) if dim > 4 else 0)
            l2b = _sse_l2(a[:body], b[:body]) if body else np.float32(0)
            ipb = _sse_ip_sum(a[:body], b[:body]) if body else np.float32(0)
            t2 = np.float32(0)
            ti = np.float32(0)
            for i in range(body, dim):
                d = np.float32(a[i] - b[i])
                t2 = np.float32(t2 + np.float32(d * d))
                ti = np.float32(ti + np.float32(a[i] * b[i]))
            el2 = np.float32(l2b + t2) if body else t2
            if body:  # hnswlib 0.5.x: both halves are distances (1 - sum), combined as res + res_tail - 1.0f
                eip = np.float32(np.float32(np.float32(np.float32(1.0) - ipb) + np.float32(np.float32(1.0) - ti))
                                 - np.float32(1.0))
            else:
                eip = np.float32(np.float32(1.0) - ti)
        assert np.float32(pyoracle.dist(METRIC_L2, a, b)).tobytes() == np.float32(el2).tobytes()
        assert np.float32(pyoracle.dist(METRIC_IP, a, b)).tobytes() == np.float32(eip).tobytes()


def test_normalize_convention():
    rng = np.random.default_rng(1)
    v = rng.standard_normal(768).astype(np.float32)
    s = np.float32(0)
    for x in v:
        s = np.float32(s + np.float32(x * x))
    inv = np.float32(np.float32(1.0) / np.float32(np.sqrt(s) + np.float32(1e-30)))
    np.testing.assert_array_equal(pyoracle.normalize(v), (v * inv).astype(np.float32))


# ---- HNSW vs exhaustive: recall sanity + work counters -------------------------------------
def test_hnsw_recall_and_counters():
    rng = np.random.default_rng(7)
    X = rng.standard_normal((2000, 32)).astype(np.float32)
    Q = rng.standard_normal((50, 32)).astype(np.float32)
    h = pyoracle.Hnsw(32, METRIC_L2, 2048)
    h.add_rows(X)
    h.set_ef(200)
    labels, dists, counts, _, st = h.search_batch(Q, 10, threads=2)
    truth, tdist, _ = pyoracle.exhaustive(X, Q, 10, METRIC_L2)
    recall = np.mean([len(set(labels[i]) & set(truth[i])) / 10 for i in range(50)])
    assert recall >= 0.95
    assert (counts == 10).all()
    # returned distances are the true distances of the returned ids
    for i in range(5):
        for j in range(10):
            assert dists[i, j] == np.float32(pyoracle.dist(METRIC_L2, Q[i], X[labels[i, j]]))
    # n_dist (rows actually fetched) <= hnswlib's looser counter + entry point evaluations
    assert 0 < st["n_dist"] <= st["metric_distance_computations"] + 50
    assert st["n_hops0"] + st["n_hops_up"] == st["metric_hops"]
    # single-query path == batch path
    l1, d1 = h.search(Q[0], 10)
    np.testing.assert_array_equal(l1, labels[0])
    np.testing.assert_array_equal(d1, dists[0])


def test_hnsw_level0_stage_matches_full_search_when_flat():
    rng = np.random.default_rng(3)
    X = rng.standard_normal((300, 16)).astype(np.float32)
    h = pyoracle.Hnsw(16, METRIC_L2, 512)
    h.add_rows(X)
    l0, lv, upper = h.export_graph()
    assert l0.shape == (300, 33) and (l0[:, 0] <= 32).all()
    assert lv.max() == h.maxlevel
    for (i, level), ids in upper.items():
        assert 1 <= level <= lv[i] and len(ids) <= 16
    np.testing.assert_array_equal(h.export_vectors(), X)
