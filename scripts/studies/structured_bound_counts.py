#!/usr/bin/env python3
"""CPU study (numpy, ~1 min): on the rows and queries of bench.py's structured leg (1 M x 768 on a 16-dim manifold + 5 %
noise, independent queries), how many rows lie within the int8 filter's bound of a query's 10th best dot product?
Answer (end of round 4): median 18 / 29 / 50 within 0.013 / 0.026 / 0.040, never more than 83 — a candidate list of 128
is ample, so a list that is too short is NOT what makes the exact engine slow on these rows in some runs."""
import time

import numpy as np

d, R, n, chunk = 768, 16, 1_000_000, 65536
A = np.random.default_rng(20250213).standard_normal((R, d)).astype(np.float32) / np.sqrt(R)


def manifold(seed, rows):  # bench.py run_structured_leg
    r = np.random.default_rng(seed)
    x = r.standard_normal((rows, R)).astype(np.float32) @ A
    x += 0.05 * r.standard_normal((rows, d)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    return np.ascontiguousarray(x, dtype=np.float32)


def main():
    Q = manifold(20250212 + 1000, 256)          # SEED_QUERY + 1000: the leg's first query batch
    t0 = time.time()
    tops = np.full((256, 10), -2.0, np.float32)
    for i0 in range(0, n, chunk):
        X = manifold(20250211 + 1 + i0 // chunk, min(chunk, n - i0))   # SEED_CORPUS + 1 + chunk index
        tops = -np.sort(-np.concatenate([tops, Q @ X.T], axis=1), axis=1)[:, :10]
    d10 = tops[:, 9]
    counts = {b: np.zeros(256, np.int64) for b in (0.013, 0.026, 0.04)}
    for i0 in range(0, n, chunk):
        X = manifold(20250211 + 1 + i0 // chunk, min(chunk, n - i0))
        D = Q @ X.T
        for b in counts:
            counts[b] += (D >= (d10[:, None] - b)).sum(axis=1)
    print("seconds", round(time.time() - t0, 1))
    print("1 - dot of the 10th neighbour: median %.4f" % float(np.median(1 - d10)))
    for b, c in counts.items():
        print("rows within %.3f of the 10th best dot: median %d, p90 %d, p99 %d, max %d; queries with > 128: %d of 256" % (
            b, np.median(c), np.percentile(c, 90), np.percentile(c, 99), c.max(), (c > 128).sum()))


if __name__ == "__main__":
    main()
