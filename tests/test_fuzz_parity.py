"""Seeded random sweep of the flat path through the C ABI against the oracle's exhaustive scan: shapes nobody picked by
hand (odd dimensions, row counts either side of the engines' gates and tile sizes, one query or a ragged handful, k up to
the paged range, fp16 rows, scaled and outlier rows, exact duplicates, rows written in several batches and re-written,
spaces sharded in-process).
Bar as everywhere: ids bit-exact, distance bytes identical, counts equal.

    python tests/test_fuzz_parity.py 400 7        # a longer sweep from the command line: 400 cases, seed 7
    python tests/test_fuzz_parity.py 300 5 --i8   # shapes the int8 engine serves (>= 16 Ki rows): every loop form of its scan
"""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # (command-line use)
from oracle import pyoracle  # noqa: E402

pytestmark = pytest.mark.gpu

METRIC_NAMES = ("l2", "ip", "cosine")


def draw_case(rng):
    """one random configuration (a dict of plain values, printable in a failure message)"""
    gate = [16384, 16385, 16383, 8192, 256, 257, 255, 1, 2, 3, 63, 64, 65, 511, 513, 4097]
    n = int(rng.choice(gate)) if rng.random() < 0.3 else int(np.exp(rng.uniform(0, np.log(60000))))
    d = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 15, 16, 17, 31, 33, 63, 64, 65, 96, 100, 127, 128, 129, 192, 255, 256, 257,
                        300, 384, 512, 640, 768, 1000, 1024, 1536])) if rng.random() < 0.7 else int(rng.integers(1, 900))
    if n * d > 24_000_000:   # keep the oracle's side of a case under a second
        n = max(1, 24_000_000 // d)
    k = int(rng.choice([1, 2, 5, 10, 10, 10, 32, 47, 48])) if rng.random() < 0.85 else int(rng.choice([49, 64, 65, 100, 200]))
    nq = int(rng.choice([1, 1, 2, 3, 7, 17, 64, 100]))
    return {"n": n, "d": d, "k": k, "nq": nq, "metric": int(rng.integers(0, 3)), "f16_rows": bool(rng.random() < 0.25),
            "scale": float(rng.choice([1.0, 1.0, 1e-3, 1e3, 37.5])), "outliers": bool(rng.random() < 0.2),
            "duplicates": bool(rng.random() < 0.2), "pieces": int(rng.choice([1, 1, 2, 3])),
            "rewrite": bool(rng.random() < 0.25), "near_queries": bool(rng.random() < 0.5),
            "shards": int(rng.choice([2, 3, 4, 8])) if rng.random() < 0.2 else 0,   # rows dealt over G shards in-process
            "data_seed": int(rng.integers(0, 2 ** 31))}


def draw_case_i8(rng):
    """a configuration the INT8 engine serves (>= 16 Ki rows, <= 2048 dims): every loop form of the scan kernel — two-stage
    tiles with compile-time slots (d <= 128), stage pairs (d = 384, 640 ...), ring revolutions (256, 512, 768 ...), odd stage
    counts (192, 320 ...), one stage (d <= 64) — one to five query tiles, chunks that own no tile, rows of varying norm under
    L2 (norm-banded tiles, B margins), structured rows, fp16 rows"""
    d = int(rng.choice([17, 48, 64, 100, 128, 129, 192, 200, 256, 320, 384, 448, 512, 640, 700, 768, 896, 1024, 1536, 2048])) \
        if rng.random() < 0.85 else int(rng.integers(8, 1100))
    nq = int(rng.choice([1, 5, 64, 200, 256, 257, 700, 1024, 1100]))
    n = int(np.exp(rng.uniform(np.log(16384), np.log(400000))))
    budget = 3.0e10   # oracle multiply-adds per case
    n = max(16384, min(n, int(budget / (d * nq))))
    if n * d * nq > 2 * budget:
        nq = max(1, int(2 * budget / (n * d)))
    k = int(rng.choice([1, 5, 10, 10, 10, 32, 48]))
    c = _draw_rest_i8(rng, n, d, k, nq)
    if c["f16_rows"] and c["scale"] > 1.0:
        c["outliers"] = False   # (50 x 37.5 x a sum over the latent dimensions leaves binary16's range)
    return c


def _draw_rest_i8(rng, n, d, k, nq):
    return {"n": n, "d": d, "k": k, "nq": nq, "metric": int(rng.integers(0, 3)), "f16_rows": bool(rng.random() < 0.2),
            "scale": float(rng.choice([1.0, 1.0, 1.0, 1e-3, 37.5])), "outliers": bool(rng.random() < 0.25),
            "duplicates": bool(rng.random() < 0.15), "pieces": int(rng.choice([1, 1, 2, 3])),
            "rewrite": bool(rng.random() < 0.2), "near_queries": bool(rng.random() < 0.5), "shards": 0,
            "latent": int(rng.choice([0, 0, 8, 16])),   # > 0: rows on a low-dimensional manifold (how stored embeddings look)
            "data_seed": int(rng.integers(0, 2 ** 31))}


def run_case(ehx, c):
    g = np.random.default_rng(c["data_seed"])
    n, d, k, nq = c["n"], c["d"], c["k"], c["nq"]
    X = (g.standard_normal((n, d)) * c["scale"]).astype(np.float32)
    if c.get("latent"):
        Z = g.standard_normal((n, c["latent"])).astype(np.float32)
        W = g.standard_normal((c["latent"], d)).astype(np.float32)
        X = ((Z @ W + np.float32(0.05) * g.standard_normal((n, d)).astype(np.float32)) * np.float32(c["scale"])).astype(np.float32)
    if c["outliers"] and n > 4:
        X[g.integers(0, n, size=max(1, n // 100))] *= np.float32(50.0)
    if c["duplicates"] and n > 8:
        src = g.integers(0, n, size=max(1, n // 10))
        X[g.integers(0, n, size=src.size)] = X[src]
    if c["metric"] == 2:   # a zero row has no direction: the reference never sees one (hnswlib normalises with +1e-30)
        X[np.abs(X).sum(axis=1) == 0, 0] = np.float32(c["scale"])
    Q = (g.standard_normal((nq, d)) * c["scale"]).astype(np.float32)
    if c["near_queries"]:
        pick = g.integers(0, n, size=nq)
        Q = X[pick] + np.float32(1e-2 * c["scale"]) * g.standard_normal((nq, d)).astype(np.float32)
    em = (ehx.METRIC_L2SQ, ehx.METRIC_IP, ehx.METRIC_COSINE)[c["metric"]]
    om = (pyoracle.METRIC_L2, pyoracle.METRIC_IP, pyoracle.METRIC_COSINE)[c["metric"]]
    s = ehx.Space.unique("fuzz", d, metric=em, dtype=ehx.DTYPE_F16 if c["f16_rows"] else ehx.DTYPE_F32,
                         shards=c.get("shards", 0))
    try:
        keys = ["r%d" % i for i in range(n)]
        cuts = sorted(set([0, n] + [int(x) for x in g.integers(0, n + 1, size=c["pieces"] - 1)]))
        for a, b in zip(cuts[:-1], cuts[1:]):
            if b > a:
                s.set_batch(keys[a:b], X[a:b])
        if c["rewrite"] and n > 2:
            s.knn(Q, min(k, 3))   # (scan copies exist before the rows change under them)
            idx = np.unique(g.integers(0, n, size=max(1, n // 20)))
            X[idx] = (g.standard_normal((idx.size, d)) * c["scale"]).astype(np.float32)
            s.set_batch([keys[i] for i in idx], X[idx])
        Xo = X.astype(np.float16).astype(np.float32) if c["f16_rows"] else X
        ids, dist, cnt = s.knn(Q, k)
        o_ids, o_dist, o_cnt = pyoracle.exhaustive(Xo, Q, k, om)
        assert np.array_equal(cnt, o_cnt), "counts differ"
        for i in range(nq):
            m = int(cnt[i])
            assert list(ids[i, :m]) == list(o_ids[i, :m]), "query %d: ids differ" % i
            assert dist[i, :m].tobytes() == o_dist[i, :m].tobytes(), "query %d: distance bytes differ" % i
        if not c.get("shards"):
            assert s.stats()["n_uncertified"] == 0
    finally:
        s.drop()


def sweep(ehx, cases, seed):
    rng = np.random.default_rng(seed)
    for j in range(cases):
        c = draw_case(rng)
        if c["f16_rows"] and c["scale"] == 1e3 and c["outliers"]:
            c["outliers"] = False   # 50 000 x N(0,1) leaves binary16's range: not a row an fp16 space can hold
        try:
            run_case(ehx, c)
        except AssertionError as e:
            raise AssertionError("case %d of seed %d: %s: %r" % (j, seed, e, c)) from None


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_seeded_random_cases_match_the_oracle(seed):
    ehx = pytest.importorskip("embeddinghub_amd")
    sweep(ehx, 40, seed)


def test_seeded_random_int8_engine_cases_match_the_oracle():
    """the same bar on shapes the int8 engine serves (draw_case_i8): `python tests/test_fuzz_parity.py 300 5 --i8` for more"""
    ehx = pytest.importorskip("embeddinghub_amd")
    rng = np.random.default_rng(606)
    for j in range(10):
        c = draw_case_i8(rng)
        try:
            run_case(ehx, c)
        except AssertionError as e:
            raise AssertionError("int8 case %d: %s: %r" % (j, e, c)) from None


if __name__ == "__main__":
    import torch  # noqa: F401
    import embeddinghub_amd as ehx_mod
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    i8_only = "--i8" in sys.argv[3:]
    rng0 = np.random.default_rng(seed0)
    failed = 0
    for j in range(n_cases):
        case = draw_case_i8(rng0) if i8_only else draw_case(rng0)
        if case["f16_rows"] and case["scale"] == 1e3 and case["outliers"]:
            case["outliers"] = False
        try:
            run_case(ehx_mod, case)
        except Exception as e:  # noqa: BLE001 - every failure is listed, the sweep goes on
            failed += 1
            print("FAIL case %d: %s: %r" % (j, e, case), flush=True)
    print("fuzz sweep%s: %d cases, seed %d, %d failed" % (" (int8 engine shapes)" if i8_only else "", n_cases, seed0, failed))
    sys.exit(1 if failed else 0)
