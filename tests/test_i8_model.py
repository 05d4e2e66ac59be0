"""CPU model of the int8 filter's score lower bound (host logic, no GPU, no oracle): the quantities make_scan8 /
prep_queries8 (k_misc.hip) store and the expressions i8_score / i8_alarm_k (k_flati8.hip) evaluate, restated in
numpy float32, checked against float64 distances on ordinary and adversarial data:

  * S_lower(r, q) mapped to a distance (D = u*S + v) never exceeds the true distance;
  * the alarm test (I * |A_r| against one threshold per tile and query) never hides a row whose S_lower is at or
    below the threshold.

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
def test_alarm_threshold_never_hides_a_hit(metric):
    """i8_alarm_k: an accumulator belongs to a row with S_lower <= thr only if I * |A_r| >= K(tile, query)"""
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
        Cmax, Dmax, Bmin = np.abs(C).max(), np.abs(D).max(), B.min()                          # tile_params8
        for qj in range(Q.shape[0]):
            col = np.sort(S[:, qj])
            for thr in (col[0], col[5], col[40], col[-1], f32(np.inf), col[0] - f32(1.0)):
                kq = _alarm_k(Bmin, Cmax, Dmax, g[qj], eq[qj], sq[qj], thr)
                prod = (I[:, qj].astype(f32) * np.abs(A)).astype(f32)                         # (float)I * |A_r|
                hits = S[:, qj] <= thr
                assert (prod[hits] >= kq).all(), (name, metric, qj, float(thr))


def _alarm_k(Bmin, Cmax, Dmax, g, eq, sq, thr):
    bg, ce = f32(Bmin * g), f32(Cmax * eq)
    num = f32(f32(f32(bg - Dmax) - ce) - thr)
    num = f32(num - f32(1e-5) * f32(abs(bg) + Dmax + ce + abs(thr))) if np.isfinite(thr) else f32(-np.inf)
    if not (num > 0):
        return f32(-np.inf)
    if not (sq > 0):
        return f32(np.inf)
    return f32(f32(num / sq) * f32(1.0 - 1e-5))


def test_the_alarm_is_judged_per_row():
    """Why the alarm compares I * |A_r| and not I against one integer per (tile, query): that integer has to assume the
    tile's largest quantisation step.  768-dim Gaussian rows, threshold near the 256th best of 10 M rows: the
    per-tile level sends nearly every 32 x 32 accumulator block down the slow path, the per-row product under a fifth
    of them — for the same candidates."""
    d, n = 768, 256 * 16
    rng = np.random.default_rng(1)
    X = rng.standard_normal((n, d)).astype(f32)
    Q = rng.standard_normal((64, d)).astype(f32)
    qi, sq, eq, g, u, v = _query_params(Q, "cosine", d)
    xi, A, B, C, D = _row_params(X, "cosine", d)
    thr = f32(1.0 - 0.1462)
    I = xi @ qi.T
    t = (sq[None, :] * I.astype(f32)).astype(f32)
    S = (A[:, None] * t + (B[:, None] * g[None, :] + (C[:, None] * eq[None, :] + D[:, None]))).astype(f32)
    per_tile = per_row = blocks = 0
    for t0 in range(0, n, 256):
        sl = slice(t0, t0 + 256)
        Amax, Cmax, Dmax, Bmin = np.abs(A[sl]).max(), np.abs(C[sl]).max(), np.abs(D[sl]).max(), B[sl].min()
        kq = np.array([_alarm_k(Bmin, Cmax, Dmax, g[j], eq[j], sq[j], thr) for j in range(Q.shape[0])], dtype=f32)
        a_row = (I[sl].astype(f32) * np.abs(A[sl])[:, None]).astype(f32) >= kq[None, :]
        a_tile = (I[sl].astype(f32) * Amax) >= kq[None, :]          # what one level per (tile, query) amounts to
        assert (a_row | ~(S[sl] <= thr)).all()                      # every candidate raises the alarm
        per_row += int(a_row.reshape(8, 32, 2, 32).any(axis=(1, 3)).sum())
        per_tile += int(a_tile.reshape(8, 32, 2, 32).any(axis=(1, 3)).sum())
        blocks += 16
    assert per_tile / blocks > 0.9 and per_row / blocks < 0.25, (per_tile / blocks, per_row / blocks)


def test_block_integer_alarm_is_valid_and_pays_once_rows_are_ordered_by_step():
    """Groundwork for the next step of the epilogue (DESIGN.md open item 1): ONE integer level per (32-row block, query),
    level = floor(K / max|A| of the block) - 1, is valid for any row order (no candidate below the level) and — once the
    rows of a tile are ordered by their step — alarms about as rarely as the per-row product does, at a third of the
    instructions.  In id order it is nearly as bad as the per-tile level."""
    d, n = 768, 256 * 8
    rng = np.random.default_rng(7)
    X = rng.standard_normal((n, d)).astype(f32)
    Q = rng.standard_normal((64, d)).astype(f32)
    qi, sq, eq, g, u, v = _query_params(Q, "cosine", d)
    xi, A, B, C, D = _row_params(X, "cosine", d)
    thr = f32(1.0 - 0.1462)
    I = xi @ qi.T
    t = (sq[None, :] * I.astype(f32)).astype(f32)
    S = (A[:, None] * t + (B[:, None] * g[None, :] + (C[:, None] * eq[None, :] + D[:, None]))).astype(f32)
    rates = {}
    for ordered in (False, True):
        alarms = blocks = 0
        for t0 in range(0, n, 256):
            idx = np.arange(t0, t0 + 256)
            if ordered:
                idx = idx[np.argsort(np.abs(A[idx]), kind="stable")]   # the per-tile permutation
            Cmax, Dmax, Bmin = np.abs(C[idx]).max(), np.abs(D[idx]).max(), B[idx].min()
            kq = np.array([_alarm_k(Bmin, Cmax, Dmax, g[j], eq[j], sq[j], thr) for j in range(64)], dtype=np.float64)
            Ablk = np.abs(A[idx]).reshape(8, 32).max(axis=1).astype(np.float64)
            level = np.floor(kq[None, :] / Ablk[:, None]) - 1.0            # integer level per (block, query)
            al = I[idx].reshape(8, 32, 64) >= level[:, None, :]
            assert (al.reshape(256, 64) | ~(S[idx] <= thr)).all()           # valid in any order
            alarms += int(al.reshape(8, 32, 2, 32).any(axis=(1, 3)).sum())
            blocks += 16
        rates[ordered] = alarms / blocks
    assert rates[False] > 0.8 and rates[True] < 0.35, rates


def _kernel_order(A, norms):
    """rank_tiles8_kernel (k_misc.hip, round 6): the positions of a FULL tile's 256 rows — pure step order when the norms
    spread by less than 0.1 %, else four norm bands of 64 rows, each by step; returns the rows in rank order (rank r sits in
    lane group r // 32)"""
    idx = np.arange(256)
    if norms.max() <= norms.min() * f32(1.001):
        return idx[np.argsort(np.abs(A), kind="stable")]
    by_norm = idx[np.argsort(norms, kind="stable")]
    return np.concatenate([b[np.argsort(np.abs(A[b]), kind="stable")] for b in (by_norm[i * 64:(i + 1) * 64] for i in range(4))])


def test_group_b_margin_is_sound_and_pays_on_rows_whose_norms_vary():
    """Round 6 (L2^2 on raw rows — the reference's default metric on un-normalised data): every 32-row lane group's alarm
    level uses the group's own min B (tileg8[tile][8 + g] = min B of the group - min B of the tile, rounded down; the
    kernel adds dB * max(gamma_q, 0) * qinv_q * (1 - 1e-4) to the tile's K) and tiles whose norms vary are ordered in four
    norm bands.  (1) SOUND on every data set of this file, any threshold: a row with S_lower <= thr always raises its
    group's alarm.  (2) It pays: on raw N(0,1) 128-dim rows the share of (group, query) pairs that alarm falls from > 0.9
    (one min B per tile, step order) to < 0.5; on normalised rows nothing changes (pure step order, margins 0)."""
    d = 128
    rng = np.random.default_rng(11)
    raw = rng.standard_normal((256 * 8, d)).astype(f32)
    unit = (raw / np.linalg.norm(raw, axis=1, keepdims=True)).astype(f32)
    sets = [("raw gaussian", raw), ("normalised", unit)] + [(n, X) for n, X in _datasets(rng, d, n=256)]
    rates = {}
    for name, X in sets:
        Q = np.concatenate([X[:6] + f32(1e-2) * rng.standard_normal((6, d)).astype(f32), rng.standard_normal((10, d)).astype(f32)])
        xi, A, B, C, D = _row_params(X, "l2", d)
        qi, sq, eq, g, u, v = _query_params(Q, "l2", d)
        norms = np.sqrt((X.astype(f32) ** 2).sum(axis=1, dtype=f32)).astype(f32)
        I = xi @ qi.T
        t = (sq[None, :] * I.astype(f32)).astype(f32)
        S = (A[:, None] * t + (B[:, None] * g[None, :] + (C[:, None] * eq[None, :] + D[:, None]).astype(f32)).astype(f32)).astype(f32)
        al_new = al_old = groups = 0
        for t0 in range(0, X.shape[0], 256):
            sl = np.arange(t0, t0 + 256)
            order = sl[_kernel_order(A[sl], norms[sl])]
            # every row's step is raised to its group's maximum (make_scan8_kernel, tgtA): |A_r| of a group = its max
            Ag = np.abs(A[order]).reshape(8, 32).max(axis=1)
            finite = np.isfinite(B[order])
            Bg = np.where(finite, B[order], f32(np.inf)).reshape(8, 32).min(axis=1).astype(f32)
            Cmax, Dmax, Bmin = np.abs(C[sl]).max(), np.abs(D[sl]).max(), B[sl].min()
            dB = np.where(np.isfinite(Bg) & np.isfinite(Bmin), ((Bg - Bmin).astype(f32) * f32(1.0 - 1e-6)).astype(f32), f32(0))
            dB = np.maximum(dB, f32(0))
            old_order = sl[np.argsort(np.abs(A[sl]), kind="stable")]
            Ag_old = np.abs(A[old_order]).reshape(8, 32).max(axis=1)
            for qj in range(Q.shape[0]):
                col = np.sort(S[sl, qj])
                qinv = f32(f32(1.0 - 1e-5) / sq[qj]) if sq[qj] > 0 else f32(np.inf)
                gq = f32(f32(max(g[qj], f32(0)) * qinv) * f32(1.0 - 1e-4))
                thr_rate = np.sort(S[:, qj])[4]      # (a level like a k'-th best of a big index: 5 rows of the whole set pass)
                for thr in (thr_rate, col[2], col[20], col[-1], f32(np.inf)):
                    kq = _alarm_k(Bmin, Cmax, Dmax, g[qj], eq[qj], sq[qj], thr)
                    for gi in range(8):
                        with np.errstate(invalid="ignore"):
                            kg = f32(dB[gi] * gq + kq)                      # fmaf(dB, gq, kq)
                        level = -2.1e9 if np.isnan(kg) else float(np.clip(np.float64(kg) / np.float64(Ag[gi]) * (1 - 2e-6), -2.1e9, 2.1e9))
                        rows = order[gi * 32:(gi + 1) * 32]
                        alarm = bool((I[rows, qj] >= int(level)).any()) if Ag[gi] > 0 else not (kq > -np.inf) or True
                        hit = bool((S[rows, qj] <= thr).any())
                        assert alarm or not hit, (name, qj, gi, float(thr))
                        if thr is thr_rate:
                            al_new += alarm
                            lo = -2.1e9 if not np.isfinite(kq) and kq < 0 else float(np.clip(np.float64(kq) / np.float64(Ag_old[gi]), -2.1e9, 2.1e9)) if Ag_old[gi] > 0 else -2.1e9
                            al_old += bool((I[old_order[gi * 32:(gi + 1) * 32], qj] >= int(lo)).any())
                            groups += 1
        rates[name] = (al_old / groups, al_new / groups)
    assert rates["raw gaussian"][0] > 0.9 and rates["raw gaussian"][1] < 0.5, rates
    assert abs(rates["normalised"][0] - rates["normalised"][1]) < 0.02, rates
