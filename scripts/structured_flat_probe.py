#!/usr/bin/env python3
"""The exact flat engine on bench.py's structured rows (32-dim manifold in 768 dims): rate per batch, which engine
answered, how many queries each stage lost (the int8 list widens when batches lose queries)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401
import embeddinghub_amd as ehx  # noqa: E402

n2, d, R, chunk, B, k = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000, 768, 32, 65536, 1024, 10
A = np.random.default_rng(20250213).standard_normal((R, d)).astype(np.float32) / np.sqrt(R)


def manifold(seed, rows):
    r = np.random.default_rng(seed)
    x = r.standard_normal((rows, R)).astype(np.float32) @ A
    x += 0.05 * r.standard_normal((rows, d)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    return np.ascontiguousarray(x, dtype=np.float32)


flat = ehx.Space("probe-flat", d, metric=ehx.METRIC_COSINE, initial_capacity=n2)
for i0 in range(0, n2, chunk):
    m = min(chunk, n2 - i0)
    flat.set_batch([b"%d" % i for i in range(i0, i0 + m)], manifold(ehx.SEED_CORPUS + 1 + i0 // chunk, m))
print("engine", flat.scan_engine(), "rows", len(flat), flush=True)
for b in range(8):
    Q = manifold(ehx.SEED_QUERY + 1000 + b, B)  # (+1000: independent of the corpus chunks' seeds SEED_CORPUS + 1 + i)
    flat.stats_reset()
    t0 = time.perf_counter()
    flat.knn(Q, k)
    dt = time.perf_counter() - t0
    st = flat.stats()
    print("batch %d: %.2f ms  i8 queries %d  i8 fallback %d  filter fallback %d  exhaustive %d  uncertified %d" % (
        b, dt * 1e3, st["n_i8_queries"], st["n_i8_fallback"], st["n_filter_fallback"], st["n_exhaustive"],
        st["n_uncertified"]), flush=True)
flat.drop()
