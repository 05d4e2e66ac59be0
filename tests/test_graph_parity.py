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
    s = ehx.Space.unique("graph", d, metric=em, mode=ehx.MODE_GRAPH, M=M, initial_capacity=n,
                         build_batch=0xFFFFFFFF)  # no GPU build: the oracle's graph is imported
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
    (4000, 64, 12, 100, (10, 300)),         # k > 48: searchKnn keeps max(ef, k) results (ef 10 -> 100)
    (2500, 32, 6, 300, (64,)),              # k spanning several 64-wide output passes
    (3000, 96, 20, 10, (50,)),              # short rows: two rows per 4-lane group (wave_group_dists), every
    (3000, 192, 20, 10, (50,)),             # paired row length once (32 and 64 above, 128 in the 10k case)
    (3000, 256, 20, 10, (50,)),
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


@pytest.mark.parametrize("em,om", METRICS)
@pytest.mark.parametrize("n,d,k,ef", [(10000, 128, 10, 10), (3000, 768, 10, 64), (2000, 50, 20, 10), (900, 19, 64, 10)])
def test_one_query_per_call_in_one_launch_walks_the_oracles_path(n, d, k, ef, em, om):
    """The reference's request shape in the reference's index (one query per NearestNeighbor RPC against the HNSW graph,
    server.cc:172-210): `ehx_knn` with one query runs the graph search as ONE launch — raw query read from host-visible
    memory and prepared by the kernel itself (cosine: normalised there), results written to host-visible memory — and
    must return what the batched path returns: the oracle's ids, distance bytes, counts and traversal counters, query
    after query (the visited bitmap is the wave's to clear), and again after a batch call in between."""
    X, h, s, rng = _build(n, d, em, om, seed=n + d + k)
    Q = rng.standard_normal((24, d)).astype(np.float32)
    h.set_ef(ef)
    s.set_ef(ef)
    labels, dists, counts, _, st = h.search_batch(Q, k, threads=1)
    s.stats_reset()
    for i in range(12):
        ids, dist, cnt = s.knn(Q[i:i + 1], k)
        assert cnt[0] == counts[i] and list(ids[0]) == list(labels[i]) and dist[0].tobytes() == dists[i].tobytes(), i
    ids, dist, cnt = s.knn(Q[12:20], k)        # a batch in between (prepare launch + search launch, memset or log)
    np.testing.assert_array_equal(ids, labels[12:20])
    assert dist.tobytes() == dists[12:20].tobytes()
    for i in range(20, 24):
        ids, dist, cnt = s.knn(Q[i:i + 1], k)
        assert cnt[0] == counts[i] and list(ids[0]) == list(labels[i]) and dist[0].tobytes() == dists[i].tobytes(), i
    g = s.stats()
    assert g["n_queries"] == 24
    assert g["n_dist"] == st["n_dist"] - 24 and g["n_hops"] == st["n_hops0"] + st["n_hops_up"]
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


# ---- GPU-side insertion (k_insert.hip): ANNIndex::set -> addPoint, index.cc:20-37 -----------------
def _same_graph(s, h):
    l0, lv, upper, ep, ml = s.graph_export()
    ol0, olv, oupper = h.export_graph()
    assert (ep, ml) == (h.enterpoint, h.maxlevel)
    np.testing.assert_array_equal(lv, olv)
    bad = np.nonzero((l0 != ol0).any(axis=1))[0]
    assert len(bad) == 0, "level-0 lists differ at nodes %s" % bad[:10]
    assert sorted(upper) == sorted(oupper)
    for key in oupper:
        np.testing.assert_array_equal(upper[key], oupper[key])


@pytest.mark.parametrize("em,om", METRICS)
@pytest.mark.parametrize("n,d", [(700, 32), (1500, 64), (400, 20)])
def test_sequential_gpu_build_is_the_oracles_graph(n, d, em, om):
    rng = np.random.default_rng(n * 7 + d)
    X = rng.standard_normal((n, d)).astype(np.float32)
    h = pyoracle.Hnsw(d, om, n)
    h.add_rows(X)
    s = ehx.Space.unique("gbuild", d, metric=em, mode=ehx.MODE_GRAPH, initial_capacity=n)
    for i in range(40):  # single Sets ...
        s.set("k%d" % i, X[i])
    s.set_batch(["k%d" % i for i in range(40, n)], X[40:])  # ... and a batch: both insert sequentially
    _same_graph(s, h)
    Q = rng.standard_normal((16, d)).astype(np.float32)
    for ef in (10, 100):
        h.set_ef(ef)
        s.set_ef(ef)
        labels, dists, counts, _, _ = h.search_batch(Q, 10, threads=1)
        ids, dist, cnt = s.knn(Q, 10)
        np.testing.assert_array_equal(ids, labels)
        assert dist.tobytes() == dists.tobytes()
    s.drop()


def test_batched_gpu_build_recall_parity_with_oracle():
    n, d, nq, k = 20000, 64, 256, 10
    X = pyoracle.gen_rows(ehx.SEED_CORPUS, 0, n, d)
    Q = pyoracle.gen_rows(ehx.SEED_QUERY, 0, nq, d)
    h = pyoracle.Hnsw(d, pyoracle.METRIC_L2, n)
    h.add_rows(X)
    s = ehx.Space.unique("gbulk", d, metric=ehx.METRIC_L2SQ, mode=ehx.MODE_GRAPH, initial_capacity=n)
    s.fill_synthetic(ehx.SEED_CORPUS, 0, n, False)  # bulk load: rounds of concurrent inserts
    truth, _, _ = pyoracle.exhaustive(X, Q, k, pyoracle.METRIC_L2)
    for ef in (50, 200):
        h.set_ef(ef)
        s.set_ef(ef)
        labels, _, _, _, _ = h.search_batch(Q, k, threads=8)
        ids, _, cnt = s.knn(Q, k)
        assert (cnt == k).all()
        r_o = np.mean([len(set(labels[i]) & set(truth[i])) / k for i in range(nq)])
        r_g = np.mean([len(set(ids[i]) & set(truth[i])) / k for i in range(nq)])
        assert r_g >= r_o - 0.02, (ef, r_g, r_o)
    l0, lv, upper, ep, ml = s.graph_export()
    assert (l0[:, 0] >= 1).all() and (l0[:, 0] <= 32).all()  # every node linked, degree bound respected
    s.drop()


def test_bulk_build_gives_its_scratch_back_and_the_space_keeps_working():
    """A bulk build releases its per-insertion visited bitmaps and link scratch when it returns (ehx_graph.cpp
    graph_insert: from 1 GiB on by default — 5 GB at 10 M rows; EHX_BUILD_SCRATCH_KEEP=0 forces it here).  Afterwards:
    searches allocate their own bitmaps, a second bulk Set allocates fresh scratch, sequential Sets and updates work —
    and every answer equals that of a space that kept its scratch (the library reads the variable once per process, so
    the comparison is against the CPU oracle's recall on the same rows and against the exact scan)."""
    import os
    import subprocess
    import sys
    code = r"""
import numpy as np
import embeddinghub_amd as ehx
from oracle import pyoracle
n, d, nq, k = 20000, 64, 128, 10
X = pyoracle.gen_rows(ehx.SEED_CORPUS, 0, n, d)
Q = pyoracle.gen_rows(ehx.SEED_QUERY, 0, nq, d)
s = ehx.Space.unique("gscratch", d, metric=ehx.METRIC_L2SQ, mode=ehx.MODE_GRAPH, initial_capacity=n + 4096, build_batch=1024)
keys = ["r%d" % i for i in range(n)]
s.set_batch(keys[:12000], X[:12000])       # bulk rounds, scratch released at the end
s.set_ef(200)
ids, dist, cnt = s.knn(Q, k)               # the search allocates its own bitmaps
assert (cnt == k).all()
s.set_batch(keys[12000:], X[12000:])       # second bulk Set: fresh scratch, released again
s.set("extra", X[5] * np.float32(0.5))     # sequential insert
s.set("r7", X[8])                          # update in place
assert np.array_equal(s.get("r7"), X[8]) and len(s) == n + 1
X2 = X.copy(); X2[7] = X[8]
X2 = np.concatenate([X2, (X[5] * np.float32(0.5))[None]])
truth, _, _ = pyoracle.exhaustive(X2, Q, k, pyoracle.METRIC_L2)
ids, dist, cnt = s.knn(Q, k)
r = np.mean([len(set(ids[i]) & set(truth[i])) / k for i in range(nq)])
l0, lv, upper, ep, ml = s.graph_export()
assert (l0[:n, 0] >= 1).all() and (l0[:, 0] <= 32).all()
print("RECALL %.4f" % r)
s.drop()
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    recalls = []
    for keep in ("0", None):
        env = dict(os.environ, PYTHONPATH=root)
        if keep is None:
            env.pop("EHX_BUILD_SCRATCH_KEEP", None)
        else:
            env["EHX_BUILD_SCRATCH_KEEP"] = keep
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600, cwd=root)
        assert out.returncode == 0, out.stderr[-2000:]
        recalls.append(float([l for l in out.stdout.splitlines() if l.startswith("RECALL")][0].split()[1]))
    assert recalls[0] >= 0.5 and abs(recalls[0] - recalls[1]) <= 0.02, recalls


@pytest.mark.parametrize("M", [8, 32])
@pytest.mark.parametrize("em,om", [(ehx.METRIC_L2SQ, pyoracle.METRIC_L2), (ehx.METRIC_COSINE, pyoracle.METRIC_COSINE)])
def test_other_degrees_M8_and_M32(M, em, om):
    """M is a reference parameter (hnswlib default 16): M=8 -> level-0 lists of 16 (one distance pass of 16
    rows), M=32 -> lists of 64 (four passes, every lane of the wave holds a neighbour).  Strict parity on the
    oracle's graph, and the sequential GPU build reproduces the oracle's graph."""
    n, d, nq, k = 2500, 48, 32, 10
    X, h, s, rng = _build(n, d, em, om, seed=M, M=M)
    Q = rng.standard_normal((nq, d)).astype(np.float32)
    for ef in (10, 150):
        h.set_ef(ef)
        s.set_ef(ef)
        s.stats_reset()
        labels, dists, counts, _, st = h.search_batch(Q, k, threads=1)
        ids, dist, cnt = s.knn(Q, k)
        np.testing.assert_array_equal(cnt, counts)
        np.testing.assert_array_equal(ids, labels)
        assert dist.tobytes() == dists.tobytes()
        g = s.stats()
        assert g["n_dist"] == st["n_dist"] - nq
        assert g["n_hops"] == st["n_hops0"] + st["n_hops_up"]
    s.drop()
    nb = 600 if M < 32 else 1500     # M = 32: enough rows that full 64-entry lists meet a 65th candidate many times
    hb = pyoracle.Hnsw(d, om, nb, M=M)
    hb.add_rows(X[:nb])
    b = ehx.Space.unique("gbuildM", d, metric=em, mode=ehx.MODE_GRAPH, M=M, initial_capacity=nb)
    b.set_batch(["k%d" % i for i in range(nb)], X[:nb])
    _same_graph(b, hb)
    b.drop()


def test_set_batch_bulk_rounds_opt_in():
    """ehx_params.build_batch > 1: an ehx_set_batch of fresh keys joins the graph in concurrent rounds
    (hnswlib-python's multi-threaded add_items, offlinehub.py:89); recall stays at the oracle's level, keys
    and Get work, and a later batch that re-writes a key falls back to the sequential update path."""
    n, d, nq, k = 12000, 48, 200, 10
    rng = np.random.default_rng(5)
    A = rng.standard_normal((12, d)).astype(np.float32)
    X = (rng.standard_normal((n, 12)).astype(np.float32) @ A + 0.05 * rng.standard_normal((n, d))).astype(np.float32)
    Q = (rng.standard_normal((nq, 12)).astype(np.float32) @ A + 0.05 * rng.standard_normal((nq, d))).astype(np.float32)
    h = pyoracle.Hnsw(d, pyoracle.METRIC_L2, n)
    h.add_rows(X)
    s = ehx.Space.unique("gset", d, metric=ehx.METRIC_L2SQ, mode=ehx.MODE_GRAPH, initial_capacity=128, build_batch=512)
    keys = ["row%d" % i for i in range(n)]
    for i0 in range(0, n, 5000):  # three calls: growth past the initial capacity, rounds inside every call
        s.set_batch(keys[i0:i0 + 5000], X[i0:i0 + 5000])
    assert len(s) == n
    truth, _, _ = pyoracle.exhaustive(X, Q, k, pyoracle.METRIC_L2)
    h.set_ef(100)
    s.set_ef(100)
    labels, _, _, _, _ = h.search_batch(Q, k, threads=8)
    ids, _, cnt = s.knn(Q, k)
    assert (cnt == k).all()
    r_o = np.mean([len(set(labels[i]) & set(truth[i])) / k for i in range(nq)])
    r_g = np.mean([len(set(ids[i]) & set(truth[i])) / k for i in range(nq)])
    assert r_o >= 0.95, r_o          # structured data: the graph index works
    assert r_g >= r_o - 0.02, (r_g, r_o)
    assert np.array_equal(s.get("row77"), X[77])
    l0, lv, upper, ep, ml = s.graph_export()
    assert (l0[:, 0] >= 1).all() and (l0[:, 0] <= 32).all()
    for row in l0[:2000]:            # lists hold distinct ids (the search kernel's visited test relies on it)
        c = int(row[0])
        assert len(set(row[1:1 + c].tolist())) == c
    # a batch that re-writes a known key goes row by row (update-in-place), the graph stays complete
    s.set_batch(["row5", "fresh"], np.stack([X[9], X[10]]))
    assert len(s) == n + 1 and np.array_equal(s.get("row5"), X[9])
    ids2, _, _ = s.knn(X[10:11], 2)
    assert set(ids2[0].tolist()) == {10, n}
    s.drop()


def test_update_in_place_repairs_the_graph_like_hnswlib():
    """index.cc:21-36: Set on an existing key keeps the label and takes hnswlib's updatePoint branch
    (re-selection of the one-hop neighbours' links among the two-hop set + repairConnectionsForUpdate)."""
    rng = np.random.default_rng(77)
    n, d = 900, 32
    X = rng.standard_normal((n, d)).astype(np.float32)
    h = pyoracle.Hnsw(d, pyoracle.METRIC_L2, n)
    s = ehx.Space.unique("gupd", d, metric=ehx.METRIC_L2SQ, mode=ehx.MODE_GRAPH, initial_capacity=n)
    h.add_rows(X)
    s.set_batch(["k%d" % i for i in range(n)], X)
    _same_graph(s, h)
    upd = [3, 500, 11, h.enterpoint, 899, 3]  # incl. the entry point and a key updated twice
    for j, i in enumerate(upd):
        v = rng.standard_normal(d).astype(np.float32)
        X[i] = v
        h.add(v, i)           # addPoint with an existing label -> updatePoint
        s.set("k%d" % i, v)
        _same_graph(s, h)
    # a batch that mixes fresh keys and re-writes, replayed in call order
    Y = rng.standard_normal((6, d)).astype(np.float32)
    keys = ["k%d" % n, "k7", "k%d" % (n + 1), "k%d" % n, "k8", "k%d" % (n + 2)]
    labels = [n, 7, n + 1, n, 8, n + 2]
    h.resize(n + 8)
    for v, lab in zip(Y, labels):
        h.add(v, lab)
    s.set_batch(keys, Y)
    _same_graph(s, h)
    Q = rng.standard_normal((16, d)).astype(np.float32)
    h.set_ef(64)
    s.set_ef(64)
    labels_o, dists_o, _, _, _ = h.search_batch(Q, 10, threads=1)
    ids, dist, _ = s.knn(Q, 10)
    np.testing.assert_array_equal(ids, labels_o)
    assert dist.tobytes() == dists_o.tobytes()
    s.drop()


def test_reference_update_test_in_graph_mode():  # index_test.cc:39-49 through the graph path
    s = ehx.Space.unique("gabc", 3, mode=ehx.MODE_GRAPH)
    s.set("a", [0, 1, 0])
    s.set("b", [1, 1, 0])
    s.set("c", [1, 0, 0])
    assert s.knn_keys([0, 1, 0], 2) == [["a", "b"]]
    s.set("a", [0, -1, 0])
    assert s.knn_keys([0, 1, 0], 1) == [["b"]]
    s.drop()


@pytest.mark.parametrize("em,om", METRICS)
def test_fp16_row_storage_in_graph_mode(em, om):
    """EHX_DTYPE_F16 graph spaces (BASELINE configs[5] rows behind a graph): rows are rounded to binary16 once, on Set;
    the search copy is made from the rounded rows, so searching, building and Get behave exactly like an fp32 space
    (and the oracle) fed the rounded rows."""
    n, d, nq, k = 1800, 72, 24, 10
    rng = np.random.default_rng(31)
    X = rng.standard_normal((n, d)).astype(np.float32)
    Xh = X.astype(np.float16).astype(np.float32)
    h = pyoracle.Hnsw(d, om, n)
    h.add_rows(Xh)
    # strict parity on the oracle's graph
    s = ehx.Space.unique("gh", d, metric=em, mode=ehx.MODE_GRAPH, initial_capacity=n, build_batch=0xFFFFFFFF,
                         dtype=ehx.DTYPE_F16)
    s.set_batch(["k%d" % i for i in range(n)], X)
    assert s.get("k7").tobytes() == Xh[7].tobytes()
    l0, lv, upper = h.export_graph()
    s.graph_import(l0, lv, upper, h.enterpoint, h.maxlevel)
    Q = rng.standard_normal((nq, d)).astype(np.float32)
    for ef in (10, 120):
        h.set_ef(ef)
        s.set_ef(ef)
        labels, dists, counts, _, _ = h.search_batch(Q, k, threads=1)
        ids, dist, cnt = s.knn(Q, k)
        np.testing.assert_array_equal(cnt, counts)
        np.testing.assert_array_equal(ids, labels)
        assert dist.tobytes() == dists.tobytes()
    s.drop()   # (the GPU build over fp16 rows: test_update_in_place_every_metric_and_row_type below)


@pytest.mark.parametrize("dtype", ["f32", "f16"])
@pytest.mark.parametrize("em,om", METRICS)
def test_update_in_place_every_metric_and_row_type(em, om, dtype):
    """updatePoint on cosine / inner-product spaces and on fp16 rows (the L2 / fp32 case has its own, longer test)"""
    n, d = 700, 40
    rng = np.random.default_rng(13)
    X = rng.standard_normal((n, d)).astype(np.float32)
    rnd = (lambda a: a.astype(np.float16).astype(np.float32)) if dtype == "f16" else (lambda a: a)
    h = pyoracle.Hnsw(d, om, n)
    h.add_rows(rnd(X))
    b = ehx.Space.unique("gupd2", d, metric=em, mode=ehx.MODE_GRAPH, initial_capacity=n,
                         dtype=ehx.DTYPE_F16 if dtype == "f16" else ehx.DTYPE_F32)
    b.set_batch(["k%d" % i for i in range(n)], X)
    _same_graph(b, h)
    for i in (5, 300, h.enterpoint):
        upd = rng.standard_normal(d).astype(np.float32)
        b.set("k%d" % i, upd)
        h.add(rnd(upd), i)   # addPoint with an existing label -> updatePoint
        assert b.get("k%d" % i).tobytes() == rnd(upd).tobytes()
        _same_graph(b, h)
    b.drop()


# ---- single-copy graph storage (round 4) ------------------------------------------------
# An fp32 graph space stores its rows ONCE, in the search copy's block order, raw; cosine rows are scaled by 1/|x| on
# the fly in the kernels.  What must hold: Get returns the bytes that were Set (vectorstore_test.go:107-113), for every
# metric, for row lengths that are not a multiple of the 16-float block, after in-place updates, and for a batch that
# writes the same key twice; and the search on the oracle's graph stays the oracle's (the tests above, which all run on
# such spaces).
@pytest.mark.parametrize("em,om", METRICS)
@pytest.mark.parametrize("d", [3, 16, 33, 100, 128, 768])
def test_single_copy_graph_space_get_is_byte_exact_and_search_is_the_oracles(d, em, om, monkeypatch):
    # (every fp32 graph space is single-copy by default; EHX_GRAPH_TWO_COPIES=1 — read when the space is created —
    # restores raw rows + search copy)
    monkeypatch.delenv("EHX_GRAPH_TWO_COPIES", raising=False)
    rng = np.random.default_rng(d * 7 + em)
    n = 600
    X = rng.standard_normal((n, d)).astype(np.float32) * np.float32(3.0)
    s = ehx.Space.unique("g1copy", d, metric=em, mode=ehx.MODE_GRAPH)
    keys = ["k%d" % i for i in range(n)]
    s.set_batch(keys[:400], X[:400])              # one contiguous batch
    for i in range(400, 420):                     # single-row Sets
        s.set(keys[i], X[i])
    s.set_batch(keys[420:], X[420:])
    for i in (0, 1, 15, 16, 399, 400, 419, 420, n - 1):
        assert s.get(keys[i]).tobytes() == X[i].tobytes(), i
    # in-place updates: scattered ids in one batch, and the same key twice in one batch (the last one wins)
    upd = rng.standard_normal((5, d)).astype(np.float32)
    s.set_batch(["k7", "k300", "k7", "k599", "k8"], upd)
    X[7], X[300], X[599], X[8] = upd[2], upd[1], upd[3], upd[4]
    for i in (6, 7, 8, 9, 299, 300, 301, 598, 599):
        assert s.get(keys[i]).tobytes() == X[i].tobytes(), i
    # the sequential GPU build is the oracle's graph (test_sequential_gpu_build_is_the_oracles_graph at larger sizes);
    # here: a fresh space, rows written once, search against the oracle's HNSW built over the same rows
    g = ehx.Space.unique("g1copy-b", d, metric=em, mode=ehx.MODE_GRAPH)
    g.set_batch(keys, X)
    h = pyoracle.Hnsw(d, om, n)
    h.add_rows(X)
    Q = rng.standard_normal((40, d)).astype(np.float32)
    for ef in (10, 64):
        g.set_ef(ef)
        h.set_ef(ef)
        ids, dist, cnt = g.knn(Q, 10)
        for i in range(Q.shape[0]):
            o_ids, o_dist = h.search(Q[i], 10)
            assert list(ids[i, :len(o_ids)]) == list(o_ids), (ef, i)
            assert dist[i, :len(o_dist)].tobytes() == o_dist.tobytes(), (ef, i)
    st = g.stats()
    assert st["n_rows"] == n
    s.drop()
    g.drop()
