#!/usr/bin/env python3
"""CPU study (the oracle's HNSW, reference defaults M = 16 / efC = 200, built with hnswlib's multi-threaded add_items):
how hard is bench.py's STRUCTURED workload for a graph index, as a function of (a) how the queries are drawn and (b) the
intrinsic dimension R of the rows (z ~ N(0, I_R) times a fixed R x 768 matrix, plus 5 % isotropic noise, normalised)?

(a) Until the end of round 3 the leg's query batch i came from seed SEED_QUERY + i = SEED_CORPUS + 1 + i — the seed of
    corpus chunk i — so every query was a corpus row's latent point under fresh noise: a near-duplicate lookup.
(b) With independent queries the ef a graph needs for recall@10 >= 0.95 grows quickly with R.

    python scripts/studies/structured_intrinsic_dim.py > profiles/r03_structured_intrinsic_dim_study.txt
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import pyoracle  # noqa: E402

d, k, nq = 768, 10, 256
SC, SQ = 20250211, 20250212
threads = min(8, os.cpu_count() or 1)


def generator(R):
    A = np.random.default_rng(20250213).standard_normal((R, d)).astype(np.float32) / np.sqrt(R)

    def manifold(seed, rows):
        r = np.random.default_rng(seed)
        x = r.standard_normal((rows, R)).astype(np.float32) @ A
        x += 0.05 * r.standard_normal((rows, d)).astype(np.float32)
        x /= np.linalg.norm(x, axis=1, keepdims=True)
        return np.ascontiguousarray(x, dtype=np.float32)
    return manifold


def curve(h, X, Q):
    truth, td, _ = pyoracle.exhaustive(X, Q, k, pyoracle.METRIC_COSINE, threads=threads)
    pts = []
    for ef in (10, 20, 40, 100, 200, 400):
        h.set_ef(ef)
        labels, _, _, _, st = h.search_batch(Q, k, threads=threads)
        rec = float(np.mean([len(set(labels[i]) & set(truth[i])) / k for i in range(Q.shape[0])]))
        pts.append("ef %d: recall %.3f (%d rows/query)" % (ef, rec, st["n_dist"] // Q.shape[0]))
    return float(np.median(td[:, 0])), pts


print("# (a) R = 32, the first 131 072 rows of the structured leg's corpus (two chunks), %d queries" % nq)
m = generator(32)
X = np.concatenate([m(SC + 1 + i, 65536) for i in range(2)])
h = pyoracle.Hnsw(d, pyoracle.METRIC_COSINE, X.shape[0])
h.add_rows_parallel(X, threads=threads)
for name, seed in (("round-3 queries (seed SEED_QUERY = a corpus chunk's seed)", SQ), ("independent queries (SEED_QUERY + 1000)", SQ + 1000)):
    d1, pts = curve(h, X, m(seed, 1024)[:nq])
    print("%s: median distance to the nearest row %.4f; %s" % (name, d1, "; ".join(pts)), flush=True)
print("# (b) independent queries, 100 000 rows, by intrinsic dimension")
for R in (8, 16, 24, 32):
    m = generator(R)
    X = m(5, 100000)
    h = pyoracle.Hnsw(d, pyoracle.METRIC_COSINE, X.shape[0])
    h.add_rows_parallel(X, threads=threads)
    d1, pts = curve(h, X, m(77, nq))
    print("R = %d: median distance to the nearest row %.3f; %s" % (R, d1, "; ".join(pts)), flush=True)
