"""Graph path at scale against the ORACLE (north_star: "IDs match the reference hnswlib CPU path at equal
recall@k").  The oracle-built graphs come from tests/golden/make_big_graphs.py (generated offline: the oracle inserts
0.3-1 k rows/s at these sizes; the files are git-ignored and travel with the repo snapshot) — a missing file skips.

  (a) the oracle's graph, IMPORTED: the engine's search must return the oracle's ids, distance bytes and counts for
      ef = 10 / 100 / 400 (the oracle's own results are stored next to the graph);
      (the fourth configuration, s16_200k768, is STRUCTURED data — rows on a 16-dim manifold + 5 % noise written
      through ehx_set_batch in chunks like bench.py's structured leg, with that leg's INDEPENDENT queries (seeds
      SEED_QUERY + 1000 + b): recall 0.74 / 0.88 / 0.965 / 0.998 / 1.0 at ef 10 / 20 / 40 / 100 / 200, so the +-0.005
      gate of (b) is tested where recall is ~0.95 and still moving.  Rounds 2-3 used man200k768, whose queries were
      corpus rows' latent points under fresh noise — a near-duplicate lookup with recall 0.97-0.99 at every ef;
      VERDICT r03 #6);
  (b) the same rows BUILT ON THE GPU in rounds of 4096 (what every >= 1 M-row number of this repo uses; hnswlib's
      multi-threaded add_items is the reference analogue, sdk/python/offlinehub.py:89): recall@10 against the exact
      answer, at equal ef, within 0.005 of the oracle-built graph's (BASELINE.md §2 gate), over 4096 queries;
  (c) exact fp32 distance ties: what the engine's (distance, id) order and libstdc++'s heap order disagree on.
"""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ehx = pytest.importorskip("embeddinghub_amd")
from oracle import pyoracle  # noqa: E402

BIG = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "_big")


def _load(name):
    path = os.path.join(BIG, name + ".npz")
    if not os.path.exists(path):
        pytest.skip("%s not generated (python tests/golden/make_big_graphs.py %s)" % (path, name))
    z = np.load(path)
    meta = json.loads(bytes(z["meta"]).decode())
    return z, meta


def _upper(z):
    off = z["upper_off"]
    ids = z["upper_ids"]
    return {(int(n), int(l)): ids[int(off[i]):int(off[i + 1])]
            for i, (n, l) in enumerate(zip(z["upper_node"], z["upper_level"]))}


def _structured(meta):
    return isinstance(meta["normalize"], str) and meta["normalize"].startswith("manifold")


def _manifold(meta):
    """bench.py's structured rows (run_structured_leg) and tests/golden/make_big_graphs.py's: z ~ N(0, I_R) times a fixed
    random R x d matrix, plus 5 % isotropic noise, normalised ("manifold": R = 32, "manifold:R")"""
    d = meta["dims"]
    R = int(meta["normalize"].split(":")[1]) if ":" in meta["normalize"] else 32
    A = np.random.default_rng(20250213).standard_normal((R, d)).astype(np.float32) / np.sqrt(R)

    def gen(seed, rows):
        r = np.random.default_rng(seed)
        x = r.standard_normal((rows, R)).astype(np.float32) @ A
        x += 0.05 * r.standard_normal((rows, d)).astype(np.float32)
        x /= np.linalg.norm(x, axis=1, keepdims=True)
        return np.ascontiguousarray(x, dtype=np.float32)
    return gen


def _fill(space, meta):
    """the configuration's rows into `space`: EHX-GAUSS-1 generated on the device, or the structured rows through
    ehx_set_batch in bench.py's chunks of 65536"""
    n = meta["rows"]
    if _structured(meta):
        gen, chunk = _manifold(meta), 65536
        for i0 in range(0, n, chunk):
            m = min(chunk, n - i0)
            space.set_batch([b"%d" % i for i in range(i0, i0 + m)], gen(ehx.SEED_CORPUS + 1 + i0 // chunk, m))
    else:
        space.fill_synthetic(ehx.SEED_CORPUS, 0, n, meta["normalize"])


def _queries(meta, nq):
    if _structured(meta):
        gen = _manifold(meta)
        # "manifold:R": independent queries (bench.py's q_seed0 = SEED_QUERY + 1000); "manifold": round 3's seeds
        q0 = ehx.SEED_QUERY + (1000 if ":" in meta["normalize"] else 0)
        return np.concatenate([gen(q0 + b, 1024) for b in range((nq + 1023) // 1024)])[:nq]
    return pyoracle.gen_rows(ehx.SEED_QUERY, 0, nq, meta["dims"], normalize=meta["normalize"])


def _recall(ids, truth):
    k = truth.shape[1]
    return float(np.mean([len(set(ids[i].tolist()) & set(truth[i].tolist())) / k for i in range(truth.shape[0])]))


@pytest.mark.parametrize("name", ["cos20k768", "cos200k768", "l2_1m128", "s16_200k768"])
def test_oracle_graph_at_scale_imported_and_gpu_built(name):
    z, meta = _load(name)
    n, d, norm = meta["rows"], meta["dims"], meta["normalize"]
    em = {pyoracle.METRIC_L2: ehx.METRIC_L2SQ, pyoracle.METRIC_IP: ehx.METRIC_IP,
          pyoracle.METRIC_COSINE: ehx.METRIC_COSINE}[meta["metric"]]
    # ---- (a) the oracle's graph, imported over rows the DEVICE generates (EHX-GAUSS-1, bit-identical to the oracle's) ----
    s = ehx.Space.unique("scale-imp", d, metric=em, mode=ehx.MODE_GRAPH, initial_capacity=n, build_batch=0xFFFFFFFF)
    _fill(s, meta)
    s.graph_import(z["level0"], z["levels"], _upper(z), int(z["entry_point"]), int(z["max_level"]))
    nq, k = meta["queries"], meta["k"]
    Q = _queries(meta, nq)
    # The structured rows come out of a float32 matmul on the host: another CPU may round them differently than the box
    # the oracle ran on.  Identity with the oracle's stored results is demanded when the rows are the same bytes (always
    # for EHX-GAUSS-1, whose generator is bit-exact everywhere); otherwise the imported graph is still the oracle's
    # graph over rows that differ in their last bits, and at most 5 % of the queries may come out differently.
    same_rows = not _structured(meta)
    if not same_rows and "rows_sha1" in meta:
        import hashlib
        gen, chunk = _manifold(meta), 65536
        hsh = hashlib.sha1()
        for i0 in range(0, n, chunk):
            hsh.update(gen(ehx.SEED_CORPUS + 1 + i0 // chunk, min(chunk, n - i0)).tobytes())
        same_rows = hsh.hexdigest() == meta["rows_sha1"]
    differing = {}
    for ef in meta["efs"]:
        s.set_ef(ef)
        ids, dist, cnt = s.knn(Q, k)
        o_ids, o_dist, o_cnt = z["ids_ef%d" % ef], z["dist_ef%d" % ef], z["cnt_ef%d" % ef]
        np.testing.assert_array_equal(cnt, o_cnt)
        bad = np.nonzero((ids != o_ids).any(axis=1) | (dist.view(np.uint32) != o_dist.view(np.uint32)).any(axis=1))[0]
        differing[ef] = len(bad)
        # bit-identical ids and distances; a query may differ only where two candidates tie EXACTLY in fp32 (heap order
        # vs (distance, id) order, test (c)) — on continuous data that is at most a query in a thousand
        limit = max(1, nq // 500) if same_rows else nq // 20
        assert len(bad) <= limit, "ef=%d: %d of %d queries differ from the oracle: %s" % (ef, len(bad), nq, bad[:8])
    # ---- (b) the same rows, built on the GPU in rounds of 4096 ----
    nq2 = 4096
    Q2 = _queries(meta, nq2)
    flat = ehx.Space.unique("scale-flat", d, metric=em, initial_capacity=n)
    _fill(flat, meta)
    truth, _, _ = flat.knn(Q2, k)                      # the exact engine (oracle-identical: tests/test_flat_parity.py)
    if same_rows:
        np.testing.assert_array_equal(truth[:nq], z["truth"])   # ... and here against the oracle's stored exhaustive scan
    flat.drop()
    g = ehx.Space.unique("scale-gpu", d, metric=em, mode=ehx.MODE_GRAPH, initial_capacity=n, build_batch=4096)
    _fill(g, meta)
    report = {"name": name, "build_div": os.environ.get("EHX_BUILD_DIV", "default"), "rows_identical_to_the_oracles": same_rows,
              "differing_queries_imported": differing, "recall": {}}
    worst = 0.0
    for ef in meta["efs"]:
        s.set_ef(ef)
        g.set_ef(ef)
        r_oracle = _recall(s.knn(Q2, k)[0], truth)
        r_gpu = _recall(g.knn(Q2, k)[0], truth)
        report["recall"][ef] = {"oracle_built": round(r_oracle, 4), "gpu_built_rounds_of_4096": round(r_gpu, 4),
                                "oracle_stored_256q": meta["search"][str(ef)]["recall_at_10"]}
        worst = max(worst, abs(r_gpu - r_oracle))
    print(json.dumps(report))
    out = os.environ.get("EHX_SCALE_REPORT")
    if out:
        with open(out, "a") as f:
            f.write(json.dumps(report) + "\n")
    s.drop()
    g.drop()
    assert worst <= 0.005, report


def test_exact_distance_ties_engine_order_vs_heap_order():
    """Rows that tie EXACTLY in fp32 (every vector stored twice): the oracle keeps hnswlib's two std::priority_queues,
    whose order among equal distances is heap layout; the engine keeps one list sorted by (distance, id).  On the
    oracle's own graph the two must still agree on every DISTANCE (the k best distances are a property of the data and
    the visited set, and with duplicates stored as mutual neighbours both twins are always reached); what may differ is
    WHICH twin's id is reported at a tied rank, and — through a different choice at the tail of the ef-list — a small
    share of the far results.  This test states that divergence: distance lists identical for >= 97 % of the queries,
    ids equal to the oracle's up to a swap of twins wherever the distances agree."""
    n_half, d, nq, k = 3000, 32, 256, 10
    rng = np.random.default_rng(21)
    base = rng.standard_normal((n_half, d)).astype(np.float32)
    X = np.concatenate([base, base])                      # id i and id i + n_half: the same vector
    Q = rng.standard_normal((nq, d)).astype(np.float32)
    n = 2 * n_half
    h = pyoracle.Hnsw(d, pyoracle.METRIC_L2, n)
    h.add_rows(X)
    s = ehx.Space.unique("ties", d, metric=ehx.METRIC_L2SQ, mode=ehx.MODE_GRAPH, initial_capacity=n,
                         build_batch=0xFFFFFFFF)
    s.set_batch(["k%d" % i for i in range(n)], X)
    l0, lv, upper = h.export_graph()
    s.graph_import(l0, lv, upper, h.enterpoint, h.maxlevel)
    for ef in (10, 50, 200):
        h.set_ef(ef)
        s.set_ef(ef)
        o_ids, o_dist, o_cnt, _, _ = h.search_batch(Q, k, threads=1)
        ids, dist, cnt = s.knn(Q, k)
        np.testing.assert_array_equal(cnt, o_cnt)
        same_dist = (dist.view(np.uint32) == o_dist.view(np.uint32)).all(axis=1)
        assert same_dist.mean() >= 0.97, (ef, float(same_dist.mean()))
        twin = lambda a: a % n_half  # noqa: E731
        for i in np.nonzero(same_dist)[0]:
            # same distances: the ids are the oracle's up to twins (sorted within each run of equal distances)
            assert sorted(zip(dist[i].tolist(), twin(ids[i]).tolist())) == sorted(zip(o_dist[i].tolist(), twin(o_ids[i]).tolist()))
        # the engine's own order inside a tie: ascending id
        for i in range(nq):
            for j in range(1, k):
                if dist[i, j] == dist[i, j - 1]:
                    assert ids[i, j] > ids[i, j - 1]
    s.drop()


def test_search_copy_of_a_fill_larger_than_one_dispatch():
    """Round-2 bug behind the stalled 10 M x 768 graph leg: the search copy was made with ONE launch of n * ld work-items;
    a dispatch holds fewer than 2^32, so a single fill of more than 2^32 elements left the tail of the copy unwritten
    (10 M x 768: rows >= 4.4 M held whatever the allocation held — in bench.py zeros, which made every later row the
    same vector, collapsed the neighbour selection onto a few hubs and turned the link kernel's per-list work serial).
    Here: one fill of 2^32 + 6 M elements, a graph in which rows on both sides of the 2^32-element mark hang off the
    entry point, and queries equal to those rows — their L2 distance through the search copy must be exactly 0."""
    d = 4096
    n = (1 << 32) // d + 1500
    s = ehx.Space.unique("bigcopy", d, metric=ehx.METRIC_L2SQ, mode=ehx.MODE_GRAPH, initial_capacity=n,
                         build_batch=0xFFFFFFFF)
    s.fill_synthetic(ehx.SEED_CORPUS, 0, n, False)
    cut = (1 << 32) // d                           # the first row the one-dispatch launch never reached
    targets = np.array([n - 1, n - 2, n - 700, cut + 7, cut, cut - 1, cut // 2, 5], dtype=np.uint32)
    ep = 100
    l0 = np.zeros((n, 33), dtype=np.uint32)        # a star: the entry point lists the rows to test, every row lists it
    l0[:, 0] = 1
    l0[:, 1] = ep
    l0[ep, 0] = len(targets)
    l0[ep, 1:1 + len(targets)] = targets
    s.graph_import(l0, np.zeros(n, dtype=np.int32), {}, ep, 0)
    s.set_ef(10)
    for row in targets.tolist():
        q = pyoracle.gen_rows(ehx.SEED_CORPUS, row, 1, d, normalize=False)
        ids, dist, cnt = s.knn(q, 1)
        assert int(ids[0, 0]) == row and float(dist[0, 0]) == 0.0, (row, ids[0, 0], dist[0, 0])
    s.drop()
