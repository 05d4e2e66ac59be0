"""CPU model of the int8 filter's score lower bound (host logic, no GPU, no oracle): the quantities make_scan8 /
prep_queries8 (k_misc.hip) store and the expression i8_score / i8_alarm_level (k_flati8.hip) evaluate, restated in
numpy float32, checked against float64 distances on ordinary and adversarial data:

  * S_lower(r, q) mapped to a distance (D = u*S + v) never exceeds the true distance;
  * the tile-level integer alarm level never hides a row whose S_lower is at or below the threshold.

The constants below are the kernels' (a change there must be mirrored here)."""
import numpy as np
import pytest

f32 = np.float32


def _slack(d):
    return f32(4e-6) + f32(1.5e-7) * f32(d)          # i8_slack


def _err_up(e2):
    return np.sqrt(e2).astype(f32) * f32(1.0 + 1e-4) + f32(3e-7)   # i8_err_up


def _quantise(V):
    """rows V (fp32) -> (n_f, ss, xi, s, e) as make_scan8 / prep_queries8 compute them"""
    ss = (V.astype(f32) ** 2).sum(axis=1, dtype=f32)
    nr = np.sqrt(ss).astype(f32)
    inv = np.where(nr > 0, f32(1) / np.where(nr > 0, nr, f32(1)), f32(0)).astype(f32)
    xh = (V * inv[:, None]).astype(f32)
    amax = np.abs(xh).max(axis=1).astype(f32)
    s = (amax / f32(127)).astype(f32)
    rs = np.where(amax > 0, f32(127) / np.where(amax > 0, amax, f32(1)), f32(0)).astype(f32)
    qf = np.clip(np.rint((xh * rs[:, None]).astype(f32)), -127, 127).astype(f32)
    res = (xh - (s[:, None] * qf).astype(f32)).astype(f32)
    e = _err_up((res ** 2).sum(axis=1, dtype=f32))
    return nr, ss, qf.astype(np.int32), s, e


def _row_params(X, metric, d):
    nr, ss, xi, s, e = _quantise(X)
    a = -np.ones_like(nr)
    b = np.ones_like(nr)
    if metric in ("ip", "l2"):
        a = -nr
    if metric == "l2":
        b = (ss * f32(1.0 - 1e-6 - 7e-8 * ((d >> 6) + 8.0))).astype(f32)
    A = (a * s).astype(f32)
    B = (b * f32(1.0 - 1e-6)).astype(f32)
    C = (a * (f32(1.0001) + e)).astype(f32)
    D = (a * (f32(1.0001) * e + _slack(d))).astype(f32)
    return xi, A, B, C, D


def _query_params(Q, metric, d):
    beta, ss, qi, s, e = _quantise(Q)
    g = np.ones_like(beta)
    u = np.ones_like(beta)
    v = np.zeros_like(beta)
    pos = beta > 0
    if metric == "ip":
        g[pos] = f32(1) / beta[pos]
        u[pos] = beta[pos]
    elif metric == "l2":
        g[pos] = f32(0.5) / beta[pos]
        u[pos] = f32(2) * beta[pos]
        v[pos] = (ss[pos] * f32(1.0 - 1e-6 - 7e-8 * ((d >> 6) + 8.0))).astype(f32)
    return qi, s, e, g, u, v


def _true_distance(X, Q, metric):
    X64, Q64 = X.astype(np.float64), Q.astype(np.float64)
    if metric == "cosine":
        nx = np.maximum(np.linalg.norm(X64, axis=1), 1e-300)
        nq = np.maximum(np.linalg.norm(Q64, axis=1), 1e-300)
        return 1.0 - (X64 / nx[:, None]) @ (Q64 / nq[:, None]).T
    if metric == "ip":
        return 1.0 - X64 @ Q64.T
    return ((X64[:, None, :] - Q64[None, :, :]) ** 2).sum(axis=2)


def _datasets(rng, d, n=320):
    g = rng.standard_normal((n, d)).astype(f32)
    yield "gaussian", g
    yield "scaled 1e3", g * f32(1e3)
    yield "scaled 1e-3", g * f32(1e-3)
    yield "near-duplicates", np.repeat(g[:16], n // 16, axis=0) + f32(1e-4) * rng.standard_normal((n, d)).astype(f32)
    sparse = np.zeros((n, d), dtype=f32)
    sparse[np.arange(n)[:, None], rng.integers(0, d, size=(n, 3))] = rng.standard_normal((n, 3)).astype(f32)
    yield "3-sparse", sparse
    yield "one-hot-ish", np.eye(d, dtype=f32)[rng.integers(0, d, n)] + f32(1e-3) * g
    yield "constant", np.ones((n, d), dtype=f32) * rng.uniform(0.5, 2, size=(n, 1)).astype(f32)
    heavy = g.copy()
    heavy[:, 0] *= f32(50)
    yield "dominant coordinate", heavy
    mixed = g * (10.0 ** rng.uniform(-1, 1, size=(n, 1))).astype(f32)
    mixed[:8] = 0
    yield "mixed norms + zero rows", mixed
    yield "all positive", np.abs(g)


@pytest.mark.parametrize("d", [8, 100, 768, 2048])
@pytest.mark.parametrize("metric", ["cosine", "ip", "l2"])
def test_lower_bound_never_exceeds_the_true_distance(d, metric):
    rng = np.random.default_rng(d)
    for name, X in _datasets(rng, d):
        Q = np.concatenate([X[:12] + f32(1e-3) * rng.standard_normal((12, d)).astype(f32),
                            rng.standard_normal((12, d)).astype(f32), X[:4]])
        xi, A, B, C, D = _row_params(X, metric, d)
        qi, sq, eq, g, u, v = _query_params(Q, metric, d)
        I = xi @ qi.T                                                       # exact integer dots [r, q]
        t = (sq[None, :] * I.astype(f32)).astype(f32)
        K = (B[:, None] * g[None, :] + (C[:, None] * eq[None, :] + D[:, None]).astype(f32)).astype(f32)
        S = (A[:, None] * t + K).astype(f32)                                # i8_score (without the fma: inside the slack)
        Dlow = (u[None, :] * S + v[None, :]).astype(f32)                    # rerank256: worst = fma(u, S, v)
        Dtrue = _true_distance(X, Q, metric)
        scale = np.maximum(np.abs(Dtrue), np.maximum((np.linalg.norm(Q.astype(np.float64), axis=1) ** 2)[None, :], 1.0))
        # the mapping's own roundings are covered by the certificate's 2e-6 * scale term (cert_margin)
        viol = (Dlow.astype(np.float64) - Dtrue - 2e-6 * scale).max()
        assert viol <= 0.0, (name, metric, d, viol)


@pytest.mark.parametrize("metric", ["cosine", "ip", "l2"])
def test_alarm_level_never_hides_a_hit(metric):
    d = 256
    rng = np.random.default_rng(3)
    for name, X in _datasets(rng, d, n=256):            # one 256-row tile
        Q = np.concatenate([X[:8] + f32(1e-2) * rng.standard_normal((8, d)).astype(f32),
                            rng.standard_normal((8, d)).astype(f32)])
        xi, A, B, C, D = _row_params(X, metric, d)
        qi, sq, eq, g, u, v = _query_params(Q, metric, d)
        I = xi @ qi.T
        t = (sq[None, :] * I.astype(f32)).astype(f32)
        K = (B[:, None] * g[None, :] + (C[:, None] * eq[None, :] + D[:, None]).astype(f32)).astype(f32)
        S = (A[:, None] * t + K).astype(f32)
        Amax, Cmax, Dmax, Bmin = np.abs(A).max(), np.abs(C).max(), np.abs(D).max(), B.min()   # tile_params8
        for qj in range(Q.shape[0]):
            col = np.sort(S[:, qj])
            for thr in (col[0], col[5], col[40], col[-1], f32(np.inf), col[0] - f32(1.0)):
                bg, ce = f32(Bmin * g[qj]), f32(Cmax * eq[qj])                                # i8_alarm_level
                num = f32(f32(f32(bg - Dmax) - ce) - thr)
                num = f32(num - f32(1e-5) * f32(abs(bg) + Dmax + ce + abs(thr))) if np.isfinite(thr) else f32(-np.inf)
                den = f32(Amax * sq[qj])
                if not (num > 0):
                    level = -2 ** 31
                elif not (den > 0):
                    level = 2 ** 31 - 1
                else:
                    lev = f32(f32(num / den) * f32(1.0 - 1e-5) - f32(1.0))
                    level = 2 ** 31 - 1 if not (lev < 2e9) else (-2 ** 31 if lev < -2e9 else int(np.floor(lev)))
                hits = S[:, qj] <= thr
                assert (I[hits, qj] >= level).all(), (name, metric, qj, float(thr))
