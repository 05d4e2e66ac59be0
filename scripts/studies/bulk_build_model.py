#!/usr/bin/env python3
"""CPU study: what does the engine's BULK graph build (rounds of rows that do not see each other; ehx_graph.cpp
graph_insert, k_insert.hip) cost in recall against hnswlib's sequential build, at equal ef?

The oracle carries a CPU MODEL of the bulk build (oracle/hnsw_oracle.hpp addPointsRounds: the same round sizes, searches
on the frozen graph, links applied in ascending id order).  Part 1 checks the model against what the GPU measured
(profiles/r03_e_graph_scale_report.jsonl: 20 k x 768 Gaussian rows, 4096 queries — GPU-built 0.045 / 0.285 / 0.668 against
oracle-built 0.046 / 0.286 / 0.671 at ef 10 / 100 / 400 with rounds of 1/256 of the graph; -0.009 at ef 400 with 1/16).
Part 2 asks the question the GPU has not answered yet (DESIGN.md §e erratum): the structured workload with INDEPENDENT
queries, where recall sits in the middle of its curve and differences between builds show.

    python scripts/studies/bulk_build_model.py > profiles/r03_bulk_build_model_study.txt
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import pyoracle  # noqa: E402

threads = min(8, os.cpu_count() or 1)
k = 10


def recalls(h, Q, truth, efs):
    out = []
    for ef in efs:
        h.set_ef(ef)
        ids, _, _, _, _ = h.search_batch(Q, k, threads=threads)
        out.append(float(np.mean([len(set(ids[i]) & set(truth[i])) / k for i in range(Q.shape[0])])))
    return out


def compare(name, X, Q, efs, divs):
    n, d = X.shape
    truth, _, _ = pyoracle.exhaustive(X, Q, k, pyoracle.METRIC_COSINE, threads=threads)
    t0 = time.time()
    s = pyoracle.Hnsw(d, pyoracle.METRIC_COSINE, n)
    s.add_rows(X)
    rs = recalls(s, Q, truth, efs)
    print("%s: sequential build (%.0f s)   recall@10 at ef %s: %s" % (name, time.time() - t0, list(efs), " / ".join("%.4f" % r for r in rs)), flush=True)
    for div in divs:
        m = pyoracle.Hnsw(d, pyoracle.METRIC_COSINE, n)
        sec = m.add_rows_rounds(X, div=div, cap=4096, threads=threads)
        rm = recalls(m, Q, truth, efs)
        print("%s: rounds of 1/%d of the graph (%.0f s): %s   difference: %s" % (
            name, div, sec, " / ".join("%.4f" % r for r in rm), " / ".join("%+.4f" % (a - b) for a, b in zip(rm, rs))), flush=True)


print("# part 1: the model against the GPU's measurements (EHX-GAUSS-1, 20 000 x 768, 4096 queries)")
X = pyoracle.gen_rows(20250211, 0, 20000, 768, normalize=True)
Q = pyoracle.gen_rows(20250212, 0, 4096, 768, normalize=True)
compare("gauss 20k", X, Q, (10, 100, 400), (256, 16))

print("# part 2: rows on a 16-dim manifold + 5 % noise, 768 dims, INDEPENDENT queries (bench.py's structured leg), 4096 queries")
A = np.random.default_rng(20250213).standard_normal((16, 768)).astype(np.float32) / np.sqrt(16)


def manifold(seed, rows):
    r = np.random.default_rng(seed)
    x = r.standard_normal((rows, 16)).astype(np.float32) @ A
    x += 0.05 * r.standard_normal((rows, 768)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    return np.ascontiguousarray(x, dtype=np.float32)


SC, SQ = 20250211, 20250212
Q = np.concatenate([manifold(SQ + 1000 + i, 1024) for i in range(4)])
for n, divs in ((100_000, (256, 128, 64)), (200_000, (128, 64))):
    X = np.concatenate([manifold(SC + 1 + i0 // 65536, min(65536, n - i0)) for i0 in range(0, n, 65536)])
    compare("structured %dk" % (n // 1000), X, Q, (10, 20, 40, 100), divs)
