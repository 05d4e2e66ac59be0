"""Seeded random sweep of graph mode through the C ABI against the oracle (restated hnswlib): random dimensions (odd ones
too), M, ef_construction, ef, k, metric, row type.  Two kinds of case:
  * import  — the oracle builds the graph, the engine imports it: every search must return the oracle's ids and
    distance bytes (searchKnn, index.cc:41);
  * build   — the engine inserts the rows one by one on the GPU (single Sets and batches, some keys re-written = hnswlib's
    updatePoint): the exported graph must BE the oracle's, and the searches as above.
Rows are continuous random values (ties between distances do not occur; tests/test_graph_scale.py states what ties do).

    python tests/test_fuzz_graph.py 300 5        # a longer sweep from the command line: 300 cases, seed 5
"""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # (command-line use)
from oracle import pyoracle  # noqa: E402

pytestmark = pytest.mark.gpu


def draw_case(rng):
    build = bool(rng.random() < 0.35)
    n = int(np.exp(rng.uniform(0, np.log(500 if build else 6000))))
    d = int(rng.choice([3, 4, 5, 7, 8, 16, 17, 20, 31, 32, 33, 64, 65, 96, 100, 127, 128, 129, 192, 256, 257, 384, 768])) \
        if rng.random() < 0.7 else int(rng.integers(3, 600))
    if n * d > 800_000:   # keep the oracle's build of a case well under a second
        n = max(1, 800_000 // d)
    M = int(rng.choice([2, 3, 4, 8, 16, 16, 16, 24, 32]))
    return {"kind": "build" if build else "import", "n": n, "d": d, "M": M,
            "efc": int(rng.choice([1, 10, 40, 200, 200, 300])), "metric": int(rng.integers(0, 3)),
            "f16_rows": bool(rng.random() < 0.2), "efs": [int(x) for x in rng.choice([1, 5, 10, 37, 64, 200, 500], size=2)],
            "k": int(rng.choice([1, 3, 10, 10, 50, 100])), "nq": int(rng.choice([1, 3, 64, 130])),
            "rewrite": bool(build and rng.random() < 0.4), "seed": int(rng.choice([100, 100, 7])),
            "data_seed": int(rng.integers(0, 2 ** 31))}


def same_graph(s, h, M):
    l0, lv, upper, ep, ml = s.graph_export()
    o0, olv, oupper = h.export_graph()
    assert ep == h.enterpoint and ml == h.maxlevel, "entry point / top level differ"
    assert np.array_equal(lv, olv), "levels differ"
    assert np.array_equal(l0[:, 0], o0[:, 0]), "level-0 degrees differ"
    for i in range(l0.shape[0]):
        c = int(o0[i, 0])
        assert np.array_equal(l0[i, 1:1 + c], o0[i, 1:1 + c]), "level-0 list of node %d differs" % i
    assert set(upper) == set(oupper), "upper lists exist for different (node, level) pairs"
    for key in oupper:
        assert np.array_equal(upper[key], oupper[key]), "upper list %r differs" % (key,)


def run_case(ehx, c):
    g = np.random.default_rng(c["data_seed"])
    n, d, k, nq, M = c["n"], c["d"], c["k"], c["nq"], c["M"]
    X = g.standard_normal((n, d)).astype(np.float32)
    if c["f16_rows"]:
        X = X.astype(np.float16).astype(np.float32)   # what an fp16 space holds; the oracle is fed the same values
    Q = g.standard_normal((nq, d)).astype(np.float32)
    em = (ehx.METRIC_L2SQ, ehx.METRIC_IP, ehx.METRIC_COSINE)[c["metric"]]
    om = (pyoracle.METRIC_L2, pyoracle.METRIC_IP, pyoracle.METRIC_COSINE)[c["metric"]]
    h = pyoracle.Hnsw(d, om, max(n, 1), M=M, ef_construction=c["efc"], seed=c["seed"])
    kw = dict(metric=em, mode=ehx.MODE_GRAPH, M=M, ef_construction=c["efc"], seed=c["seed"], initial_capacity=n,
              dtype=ehx.DTYPE_F16 if c["f16_rows"] else ehx.DTYPE_F32)
    keys = ["g%d" % i for i in range(n)]
    if c["kind"] == "import":
        h.add_rows(X)
        s = ehx.Space.unique("fzg", d, build_batch=0xFFFFFFFF, **kw)
    else:
        s = ehx.Space.unique("fzg", d, **kw)
    try:
        if c["kind"] == "import":
            s.set_batch(keys, X)
            l0, lv, upper = h.export_graph()
            s.graph_import(l0, lv, upper, h.enterpoint, h.maxlevel)
        else:
            cut = int(g.integers(0, n + 1))
            for i in range(min(cut, 25)):            # single Sets ...
                s.set(keys[i], X[i])
                h.add(X[i], i)
            a = min(cut, 25)
            if n > a:                                 # ... and a batch: both insert one row per round
                s.set_batch(keys[a:], X[a:])
                for i in range(a, n):
                    h.add(X[i], i)
            if c["rewrite"] and n > 3:
                idx = np.unique(g.integers(0, n, size=max(1, n // 15)))
                Xn = g.standard_normal((idx.size, d)).astype(np.float32)
                if c["f16_rows"]:
                    Xn = Xn.astype(np.float16).astype(np.float32)
                for j, i in enumerate(idx):           # a known key: hnswlib's updatePoint, in call order
                    X[i] = Xn[j]
                    h.add(Xn[j], int(i))
                s.set_batch([keys[i] for i in idx], Xn)
            same_graph(s, h, M)
        for ef in c["efs"]:
            h.set_ef(ef)
            s.set_ef(ef)
            o_ids, o_dist, o_cnt, _, _ = h.search_batch(Q, k, threads=1)
            ids, dist, cnt = s.knn(Q, k)
            assert np.array_equal(cnt, o_cnt), "ef=%d: counts differ" % ef
            for i in range(nq):
                m = int(cnt[i])
                assert list(ids[i, :m]) == list(o_ids[i, :m]), "ef=%d query %d: ids differ" % (ef, i)
                assert dist[i, :m].tobytes() == o_dist[i, :m].tobytes(), "ef=%d query %d: distance bytes differ" % (ef, i)
    finally:
        s.drop()


def sweep(ehx, cases, seed, report=None):
    rng = np.random.default_rng(seed)
    failed = 0
    for j in range(cases):
        c = draw_case(rng)
        try:
            run_case(ehx, c)
        except Exception as e:  # noqa: BLE001 - the case is part of the message
            if report is None:
                raise AssertionError("case %d of seed %d: %s: %r" % (j, seed, e, c)) from None
            failed += 1
            report("FAIL case %d: %s: %r" % (j, e, c))
    return failed


@pytest.mark.parametrize("seed", [1, 2])
def test_seeded_random_graph_cases_match_the_oracle(seed):
    ehx = pytest.importorskip("embeddinghub_amd")
    sweep(ehx, 25, seed)


if __name__ == "__main__":
    import torch  # noqa: F401
    import embeddinghub_amd as ehx_mod
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    bad = sweep(ehx_mod, n_cases, seed0, report=lambda m: print(m[:700], flush=True))
    print("graph fuzz sweep: %d cases, seed %d, %d failed" % (n_cases, seed0, bad))
    sys.exit(1 if bad else 0)
