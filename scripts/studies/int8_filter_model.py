#!/usr/bin/env python3
"""Reference model (CPU, numpy) of the INT8 filter planned for the flat path (flat_scan16 section):
the exact per-row / per-query quantities, the score lower bound the kernel's epilogue would form IN FLOAT32,
and a check on ordinary and adversarial data that the bound never exceeds the true distance (float64).

Scan copy of a row x (norm n_r = |x|, x^ = x / n_r):
    s_x = max|x^_i| / 127          xi = clamp(rint(x^_i / s_x), -127, 127)   (int8)
    e_r  >= |x^ - s_x xi|          (float32, rounded UP by a relative margin)
Query q (beta = |q|, q^ = q / beta): s_q, qi, e_q likewise, n8_q >= |s_q qi|.
Exact integer dot A = <qi, xi> (int32 accumulation is exact: |A| <= 127^2 * d < 2^31 for d <= 133 000).
    dot^ = <q^, x^>  in  [s_q s_x A - B, s_q s_x A + B],   B = e_q * |x8| + n8_q * e_r,   |x8| <= 1 + e_r
Every metric is affine in dot^ (k_flat16.hip):  S = b_r*gamma_q + a_r*dot^,  D = u_q*S + v_q,  a_r <= 0, u_q > 0:
    cosine  a=-1    b=1      gamma=1           u=1       v=0
    IP      a=-n_r  b=1      gamma=1/beta      u=beta    v=0       (D = 1 - <q,x>)
    L2^2    a=-n_r  b=n_r^2  gamma=1/(2 beta)  u=2 beta  v=beta^2
so  S_low = b_r*gamma_q + a_r*(s_q s_x A + e_q (1 + e_r) + n8_q e_r)  <=  S   and   D_low = u_q*S_low + v_q <= D,
formed from the row parameters (a_r s_x, a_r, a_r e_r, b_r) and the query parameters (s_q, e_q, n8_q, gamma_q).
The float32 evaluation adds rounding of a handful of operations; REL below is the relative safety margin folded
into e_r / e_q / n8_q, ABS the absolute slack subtracted from S_low (both far below the bound itself, ~1e-2)."""
import numpy as np

f32 = np.float32
REL = f32(1e-5)   # relative inflation of the stored error norms (covers their own float32 rounding)
ABS = f32(4e-6)   # absolute slack on S_low per unit of |a_r| (covers the float32 evaluation of the epilogue)


def quantise(v):
    """v: [n, d] float32 unit rows -> (vi int8, s, e_up, n8_up) with float32 arithmetic as a kernel would."""
    s = (np.abs(v).max(axis=1) / f32(127)).astype(f32)
    s = np.where(s > 0, s, f32(1)).astype(f32)
    vi = np.clip(np.rint(v / s[:, None]), -127, 127).astype(np.int8)
    v8 = (s[:, None] * vi.astype(f32)).astype(f32)
    e = np.sqrt(((v - v8).astype(f32) ** 2).sum(axis=1, dtype=f32)).astype(f32)
    n8 = np.sqrt((v8 ** 2).sum(axis=1, dtype=f32)).astype(f32)
    return vi, s, (e * (f32(1) + REL) + f32(1e-7)).astype(f32), (n8 * (f32(1) + REL)).astype(f32)


def lower_bounds(X, Q, metric):
    """D_low[q, r] in float32 (the model of the kernel) and the true D[q, r] in float64."""
    n_r = np.linalg.norm(X.astype(np.float64), axis=1)
    beta = np.linalg.norm(Q.astype(np.float64), axis=1)
    xh = (X / np.maximum(n_r, 1e-30)[:, None]).astype(f32)
    qh = (Q / np.maximum(beta, 1e-30)[:, None]).astype(f32)
    xi, s_x, e_r, _ = quantise(xh)
    qi, s_q, e_q, n8_q = quantise(qh)
    A = qi.astype(np.int32) @ xi.astype(np.int32).T                       # exact
    nr32, b32 = n_r.astype(f32), beta.astype(f32)
    if metric == "cosine":
        a_r, b_r = -np.ones_like(nr32), np.ones_like(nr32)
        gam, u, v = np.ones_like(b32), np.ones_like(b32), np.zeros_like(b32)
        D = 1.0 - (X.astype(np.float64) / n_r[:, None]) @ (Q.astype(np.float64) / beta[:, None]).T
    elif metric == "ip":
        a_r, b_r = -nr32 * (f32(1) + REL), np.ones_like(nr32)
        gam, u, v = (f32(1) / b32), b32, np.zeros_like(b32)
        D = 1.0 - X.astype(np.float64) @ Q.astype(np.float64).T
    else:
        a_r, b_r = -nr32 * (f32(1) + REL), (nr32 * nr32 * (f32(1) - REL)).astype(f32)
        gam, u, v = (f32(1) / (f32(2) * b32)), f32(2) * b32, (b32 * b32 * (f32(1) - REL)).astype(f32)
        D = ((X.astype(np.float64)[:, None, :] - Q.astype(np.float64)[None, :, :]) ** 2).sum(axis=2)
    # float32 epilogue: P1 = a_r s_x, P2 = a_r, P3 = a_r e_r
    P1, P2, P3 = (a_r * s_x).astype(f32), a_r, (a_r * e_r).astype(f32)
    sqA = (s_q[:, None] * A.astype(f32)).astype(f32)                       # [q, r]
    S_low = (b_r[None, :] * gam[:, None] + P1[None, :] * sqA + P2[None, :] * (e_q * (f32(1) + e_r.max()))[:, None]
             + P3[None, :] * n8_q[:, None]).astype(f32)
    S_low = (S_low + ABS * P2[None, :] * (f32(1) + f32(0) * S_low)).astype(f32)   # P2 <= 0: subtracts the slack
    D_low = (u[:, None] * S_low + v[:, None]).astype(f32)
    # slack of the last affine map (u, v rounded): one more relative margin on the result's magnitude
    D_low = (D_low - REL * (np.abs(D_low) + u[:, None] * np.abs(S_low) + np.abs(v)[:, None])).astype(f32)
    return D_low.T, D                                                       # both [r, q]


def datasets(rng, d):
    g = rng.standard_normal((400, d)).astype(f32)
    yield "gaussian", g
    yield "scaled 1e3", g * f32(1e3)
    yield "scaled 1e-3", g * f32(1e-3)
    near = np.repeat(g[:20], 20, axis=0) + f32(1e-4) * rng.standard_normal((400, d)).astype(f32)
    yield "near-duplicates", near
    sparse = np.zeros((400, d), dtype=f32)
    idx = rng.integers(0, d, size=(400, 3))
    sparse[np.arange(400)[:, None], idx] = rng.standard_normal((400, 3)).astype(f32)
    yield "3-sparse rows", sparse
    yield "one-hot-ish", np.eye(d, dtype=f32)[rng.integers(0, d, 400)] + f32(1e-3) * g
    yield "constant rows", np.ones((400, d), dtype=f32) * rng.uniform(0.5, 2, size=(400, 1)).astype(f32)
    heavy = g.copy()
    heavy[:, 0] *= f32(50)
    yield "one dominant coordinate", heavy


def main():
    rng = np.random.default_rng(0)
    worst = 0.0
    for d in (8, 100, 768):
        for name, X in datasets(rng, d):
            Q = np.concatenate([X[:16] + f32(1e-3) * rng.standard_normal((16, d)).astype(f32),
                                rng.standard_normal((16, d)).astype(f32)])
            for metric in ("cosine", "ip", "l2"):
                D_low, D = lower_bounds(X, Q, metric)
                viol = (D_low.astype(np.float64) - D).max()
                gap = np.median(D - D_low.astype(np.float64))
                worst = max(worst, viol)
                assert viol <= 0.0, (d, name, metric, viol)
                print("d=%4d %-24s %-6s max(D_low - D) = %+.3e   median slack %.3e" % (d, name, metric, viol, gap))
    print("the float32 lower bound never exceeded the true distance (worst %+.3e)" % worst)


if __name__ == "__main__":
    main()
