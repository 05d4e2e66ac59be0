"""Host-side check of the graph-mode SEARCH COPY layout (k_misc.hip::make_search_copy_kernel,
ehx_kernels.h::search_copy_pos / canon_dist_group_t): restated in numpy float32 — every 16-float block
permuted so that the four inputs of SSE partial sum j are contiguous, lane j of a 4-lane group adding its
four products per block in order, ((p0+p1)+p2)+p3, scalar tail last — it must reproduce the oracle's
(hnswlib SSE-order) distance bit for bit for every distance-function variant hnswlib dispatches on
(d%16==0, d%4==0, residual-16, residual-4, scalar).  The GPU kernel itself is checked against the oracle by
tests/test_graph_parity.py; this pins the layout the two sides agree on."""
import numpy as np
import pytest

from oracle import pyoracle

f32 = np.float32


def search_copy_pos(m):
    return (m & ~15) + ((m & 3) << 2) + ((m >> 2) & 3)


def make_search_copy(x, ld):
    xs = np.zeros(ld, dtype=f32)
    for m in range(len(x)):
        xs[search_copy_pos(m)] = x[m]
    return xs


def group_distance(metric, qp, xs, dims):
    if dims % 16 == 0 or dims % 4 == 0:
        body = dims
    elif dims > 16:
        body = dims & ~15
    elif dims > 4:
        body = dims & ~3
    else:
        body = 0
    p = [f32(0)] * 4
    n16, rem4 = body // 16, (body % 16) // 4
    for sub in range(4):
        acc = f32(0)
        for t in range(n16 + (1 if rem4 else 0)):
            ncomp = 4 if t < n16 else rem4
            for c in range(ncomp):
                a, b = qp[16 * t + 4 * sub + c], xs[16 * t + 4 * sub + c]
                if metric == pyoracle.METRIC_L2:
                    dlt = f32(a - b)
                    acc = f32(acc + f32(dlt * dlt))
                else:
                    acc = f32(acc + f32(a * b))
        p[sub] = acc
    res = f32(f32(f32(p[0] + p[1]) + p[2]) + p[3])
    if body != dims:
        tail = f32(0)
        for m in range(body, dims):
            a, b = qp[search_copy_pos(m)], xs[search_copy_pos(m)]
            if metric == pyoracle.METRIC_L2:
                dlt = f32(a - b)
                tail = f32(tail + f32(dlt * dlt))
            else:
                tail = f32(tail + f32(a * b))
        if metric != pyoracle.METRIC_L2 and body:
            # hnswlib 0.5.x residual variants: both halves are distances (1 - sum), combined as res + res_tail - 1
            return f32(f32(f32(f32(1) - res) + f32(f32(1) - tail)) - f32(1))
        res = f32(res + tail) if body else tail
    return f32(f32(1) - res) if metric != pyoracle.METRIC_L2 else res


def test_permutation_is_a_bijection_on_every_block():
    for blk in range(0, 64, 16):
        assert sorted(search_copy_pos(m) for m in range(blk, blk + 16)) == list(range(blk, blk + 16))
    # the four inputs of SSE partial sum j inside a block are contiguous and in order
    for j in range(4):
        assert [search_copy_pos(4 * i + j) for i in range(4)] == [4 * j + i for i in range(4)]


@pytest.mark.parametrize("dims", [3, 4, 7, 16, 19, 20, 33, 100, 128, 131, 768])
@pytest.mark.parametrize("metric", [pyoracle.METRIC_L2, pyoracle.METRIC_IP])
def test_group_arithmetic_over_the_search_copy_equals_the_oracle(dims, metric):
    rng = np.random.default_rng(dims * 7 + metric)
    ld = (dims + 31) // 32 * 32
    for _ in range(5):
        q = rng.standard_normal(dims).astype(f32)
        x = rng.standard_normal(dims).astype(f32)
        want = f32(pyoracle.dist(metric, q, x))
        got = group_distance(metric, make_search_copy(q, ld), make_search_copy(x, ld), dims)
        assert got.tobytes() == want.tobytes(), (dims, metric, got, want)


def test_cosine_rows_are_stored_normalised_like_hnswlib_python():
    # cosine = inner product over rows normalised on add (one rounding per element): the search copy stores
    # exactly that product, so the group arithmetic sees the same operands as the oracle's cosine distance
    rng = np.random.default_rng(11)
    dims, ld = 768, 768
    q = pyoracle.normalize(rng.standard_normal(dims).astype(f32))
    x = rng.standard_normal(dims).astype(f32)
    xn = pyoracle.normalize(x)
    want = f32(pyoracle.dist(pyoracle.METRIC_IP, q, xn))
    got = group_distance(pyoracle.METRIC_IP, make_search_copy(q, ld), make_search_copy(xn, ld), dims)
    assert got.tobytes() == want.tobytes()


def test_staged_rerank_slots_are_consistent_and_conflict_free():
    """wave_rows_dist_staged (k_select.hip): the wave loads 64 rows x 128 bytes together — instruction i, lane l takes
    row i*8 + l//8, physical piece l%8, which holds logical piece (l%8) ^ ((row >> 1) & 7) — parks them at
    stage[i*64 + l] and lane r reads logical piece p of its own row back from stage[r*8 + (p ^ ((r >> 1) & 7))].
    Host-side restatement of that index arithmetic: every (row, piece) comes back from where it was put, and inside
    each ds_read_b128 lane group (MI355X_MICROARCH.md, LDS table) the 16 reads touch 64 distinct banks."""
    stage = {}
    for i in range(8):
        for l in range(64):
            row, sub = i * 8 + l // 8, l % 8
            stage[i * 64 + l] = (row, sub ^ ((row >> 1) & 7))          # (row, logical piece) stored in this slot
    for r in range(64):
        for p in range(8):
            assert stage[r * 8 + (p ^ ((r >> 1) & 7))] == (r, p)
    groups = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
              list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
              list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)),
              list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]
    for p in range(8):
        for g in groups:
            banks = set()
            for lane in g:
                addr = (lane * 8 + (p ^ ((lane >> 1) & 7))) * 16     # byte address of the 16-byte read
                for b in range(4):
                    bank = (addr // 4 + b) % 64
                    assert bank not in banks, (p, lane)
                    banks.add(bank)
            assert len(banks) == 64
