#!/usr/bin/env python3
"""CPU prediction of bench.py's structured leg (1 M x 768 rows on a 16-dim manifold, independent queries): the oracle's
MODEL of the engine's bulk build (addPointsRounds) builds the index, the oracle searches it — recall-vs-ef and rows
fetched per query to hold the GPU's first measurement of the corrected leg against (erratum of section e).

    python scripts/studies/structured_leg_prediction.py > profiles/r03_structured_leg_prediction_cpu_model.txt   (~15 min, 9 GB)
"""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import pyoracle
A = np.random.default_rng(20250213).standard_normal((16, 768)).astype(np.float32) / np.sqrt(16)
def manifold(seed, rows):
    r = np.random.default_rng(seed)
    x = r.standard_normal((rows, 16)).astype(np.float32) @ A
    x += 0.05 * r.standard_normal((rows, 768)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    return np.ascontiguousarray(x, dtype=np.float32)
SC, SQ = 20250211, 20250212
n, k = 1_000_000, 10
X = np.concatenate([manifold(SC + 1 + i0 // 65536, min(65536, n - i0)) for i0 in range(0, n, 65536)])
Q = np.concatenate([manifold(SQ + 1000 + i, 1024) for i in range(4)])
truth, _, _ = pyoracle.exhaustive(X, Q, k, pyoracle.METRIC_COSINE, threads=8)
m = pyoracle.Hnsw(768, pyoracle.METRIC_COSINE, n)
sec = m.add_rows_rounds(X, div=128, cap=4096, threads=8)
out = []
for ef in (10, 20, 40, 60, 100, 200):
    m.set_ef(ef)
    ids, _, _, _, st = m.search_batch(Q, k, threads=8)
    out.append("ef %d: %.4f (%d rows/query)" % (ef, float(np.mean([len(set(ids[i]) & set(truth[i])) / k for i in range(Q.shape[0])])), st["n_dist"] // Q.shape[0]))
print("# bulk-build MODEL (rounds of 1/128 of the graph, at most 4096 rows) over bench.py's structured leg: 1 000 000 x 768, R = 16, 4096 independent queries; model build %.0f s on 8 threads" % sec)
print("recall@10: " + "; ".join(out))
