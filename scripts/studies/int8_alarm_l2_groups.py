#!/usr/bin/env python3
"""Study (CPU, numpy; the model of tests/test_i8_model.py): how often does the int8 scan's block alarm fire on L2 rows whose
NORMS vary (raw N(0,1) rows — SURVEY's C4 workload — and EHX-MANIFOLD-1-like rows), with
  (a) the shipped level: rows of a tile ordered by quantisation step |A_r|, ONE min B per tile;
  (b) rows ordered by NORM (L2: B_r = |x_r|^2 and |A_r| = |x_r| s_r both follow it), min B per 32-row lane group.
An alarm = a (32-row group, query) whose integer maximum reaches the level; every alarm costs the epilogue's slow path."""
import sys
import os
import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
from test_i8_model import _row_params, _query_params, _alarm_k, f32  # noqa: E402


def study(name, X, Q, frac):
    d = X.shape[1]
    xi, A, B, C, D = _row_params(X, "l2", d)
    qi, sq, eq, g, u, v = _query_params(Q, "l2", d)
    I = xi @ qi.T
    t = (sq[None, :] * I.astype(f32)).astype(f32)
    S = (A[:, None] * t + (B[:, None] * g[None, :] + (C[:, None] * eq[None, :] + D[:, None]))).astype(f32)
    thr = np.quantile(S, frac, axis=0).astype(f32)          # per query: the level a k'-th best of a big index sits at
    norms = np.sqrt((X.astype(np.float64) ** 2).sum(axis=1))
    n = X.shape[0]
    res = {}
    for mode in ("by step, tile min B", "by norm, group min B", "by step, group min B", "4 norm bands x 2 step halves, group min B",
                 "2 norm bands x 4 step quarters, group min B"):
        alarms = hits = groups = 0
        for t0 in range(0, n, 256):
            idx = np.arange(t0, t0 + 256)
            key = norms[idx] if mode.startswith("by norm") else np.abs(A[idx])
            if mode.startswith("by level key"):
                # the integer level a row needs for a TYPICAL query (|q| = the tile's mean norm, its k'-th neighbour z sigma out):
                # (gamma0 (B_r - Bbar) + nbar z / sqrt(d)) / |A_r| — rows that need similar levels share a group; descending
                nb = norms[idx].mean()
                key = -((0.5 * (norms[idx] ** 2 / nb - nb) + nb * 3.8 / np.sqrt(d)) / np.abs(A[idx]))
            idx = idx[np.argsort(key, kind="stable")]
            if "bands" in mode:
                nbands = int(mode.split()[0])
                idx = np.arange(t0, t0 + 256)
                idx = idx[np.argsort(norms[idx], kind="stable")]
                per = 256 // nbands
                idx = np.concatenate([b[np.argsort(np.abs(A[b]), kind="stable")] for b in (idx[i * per:(i + 1) * per] for i in range(nbands))])
            Cmax, Dmax, Bmin = np.abs(C[idx]).max(), np.abs(D[idx]).max(), B[idx].min()
            Ag = np.abs(A[idx]).reshape(8, 32).max(axis=1).astype(np.float64)      # the group's |A| (rows' steps raised to it)
            Bg = B[idx].reshape(8, 32).min(axis=1)
            for qj in range(Q.shape[0]):
                for gi in range(8):
                    bmin = Bg[gi] if "group min B" in mode else Bmin
                    kq = float(_alarm_k(bmin, Cmax, Dmax, g[qj], eq[qj], sq[qj], thr[qj]))
                    level = np.floor(kq / Ag[gi]) - 1.0 if np.isfinite(kq) else kq
                    rows = idx[gi * 32:(gi + 1) * 32]
                    al = bool((I[rows, qj] >= level).any())
                    hit = bool((S[rows, qj] <= thr[qj]).any())
                    assert al or not hit
                    alarms += al
                    hits += hit
                    groups += 1
        res[mode] = (alarms / groups, hits / groups)
    print(name, {k: "alarm %.3f (true hit %.4f)" % v for k, v in res.items()})


def main():
    rng = np.random.default_rng(5)
    d, n, nq = 128, 256 * 24, 48
    Xg = rng.standard_normal((n, d)).astype(f32)
    Qg = rng.standard_normal((nq, d)).astype(f32)
    study("raw gaussian 128", Xg, Qg, 2e-3)
    Xn = Xg / np.linalg.norm(Xg, axis=1, keepdims=True)
    study("normalised gaussian 128 (what bench.py's L2 leg used)", Xn.astype(f32), (Qg / np.linalg.norm(Qg, axis=1, keepdims=True)).astype(f32), 2e-3)
    R = 16
    Aa = rng.standard_normal((R, d)).astype(f32) / np.sqrt(R)
    Xm = (rng.standard_normal((n, R)).astype(f32) @ Aa + 0.05 * rng.standard_normal((n, d)).astype(f32)).astype(f32)
    Qm = (rng.standard_normal((nq, R)).astype(f32) @ Aa + 0.05 * rng.standard_normal((nq, d)).astype(f32)).astype(f32)
    study("manifold-16 in 128", Xm, Qm, 2e-3)


if __name__ == "__main__":
    main()
