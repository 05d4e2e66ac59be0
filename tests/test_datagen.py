"""EHX-GAUSS-1 (include/ehx_datagen.h): Philox known answers, oracle vs device bit-equality."""
import ctypes as C

import numpy as np
import pytest

from oracle import pyoracle


def test_philox_known_answers():
    # Random123 kat_vectors for philox4x32-10
    assert [hex(x) for x in pyoracle.philox([0, 0, 0, 0], [0, 0])] == \
        ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]
    assert [hex(x) for x in pyoracle.philox([0xffffffff] * 4, [0xffffffff] * 2)] == \
        ["0x408f276d", "0x41c83b0e", "0xa20bc7c6", "0x6d5451fd"]
    assert [hex(x) for x in pyoracle.philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344],
                                            [0xa4093822, 0x299f31d0])] == \
        ["0xd16cfe09", "0x94fdcceb", "0x5001e420", "0x24126ea1"]


def test_oracle_generator_is_standard_normal():
    x = pyoracle.gen_rows(20250211, 0, 400, 768)
    assert abs(float(x.mean())) < 0.01 and abs(float(x.std()) - 1.0) < 0.01
    assert abs(float((x ** 3).mean())) < 0.03 and abs(float((x ** 4).mean()) - 3.0) < 0.1
    # rows/columns are decorrelated and seeds differ
    assert abs(float(np.corrcoef(x[0], x[1])[0, 1])) < 0.15
    y = pyoracle.gen_rows(20250212, 0, 4, 768)
    assert not np.array_equal(x[:4], y)
    # normalised rows are unit length, and row r does not depend on how many rows were asked for
    n = pyoracle.gen_rows(20250211, 7, 3, 768, normalize=True)
    np.testing.assert_allclose(np.linalg.norm(n.astype(np.float64), axis=1), 1.0, atol=1e-6)
    np.testing.assert_array_equal(pyoracle.gen_rows(20250211, 8, 1, 768, normalize=True)[0], n[1])


@pytest.mark.gpu
@pytest.mark.parametrize("dims,normalize", [(768, False), (768, True), (128, True), (20, False), (1536, True)])
def test_device_generator_matches_oracle_bit_for_bit(dims, normalize):
    import torch
    from embeddinghub_amd import _lib
    L = _lib.load()
    n, row0 = 517, 123456789012
    out = torch.empty((n, dims), dtype=torch.float32, device="cuda")
    _lib.check(L.ehx_gen_rows_device(C.c_void_p(torch.cuda.current_stream().cuda_stream), 20250211, row0, n,
                                     dims, int(normalize), C.c_void_p(out.data_ptr())))
    torch.cuda.synchronize()
    ref = pyoracle.gen_rows(20250211, row0, n, dims, normalize=normalize)
    assert out.cpu().numpy().tobytes() == ref.tobytes()


@pytest.mark.gpu
def test_fill_synthetic_equals_set_of_oracle_rows():
    import embeddinghub_amd as ehx
    d, n = 64, 3000
    a = ehx.Space.unique("syn", d, metric=ehx.METRIC_COSINE)
    a.fill_synthetic(ehx.SEED_CORPUS, 0, n, True)
    X = pyoracle.gen_rows(ehx.SEED_CORPUS, 0, n, d, normalize=True)
    np.testing.assert_array_equal(a.get_by_id(n - 1), X[n - 1])
    assert a.key_of(42) == "42"
    Q = pyoracle.gen_rows(ehx.SEED_QUERY, 0, 33, d, normalize=True)
    ids, dist, cnt = a.knn(Q, 10)
    oids, odist, _ = pyoracle.exhaustive(X, Q, 10, pyoracle.METRIC_COSINE)
    np.testing.assert_array_equal(ids, oids)
    assert dist.tobytes() == odist.tobytes()


def test_oracle_manifold_rows_have_the_structure_the_spec_states():
    """EHX-MANIFOLD-1 (include/ehx_datagen.h), the oracle's restatement: rows lie near an R-dimensional linear subspace (the
    R leading singular values carry all but the 5 % noise), seeds and rows are independent draws, a row does not depend on
    the call that produced it, and R = dims-worth of structure is what makes a graph index work on them."""
    d, R, n = 256, 16, 1200
    x = pyoracle.gen_manifold_rows(20250211, 0, n, d, R)
    sv = np.linalg.svd(x.astype(np.float64), compute_uv=False)
    assert sv[R - 1] > 8 * sv[R], (sv[R - 1], sv[R])             # a sharp drop after R directions
    noise_energy = float((sv[R:] ** 2).sum() / (sv ** 2).sum())
    assert 0.001 < noise_energy < 0.01                            # 0.05^2 * (d - R) / (1 + 0.05^2 d) ~ 0.0025 * 240 / 256
    np.testing.assert_array_equal(pyoracle.gen_manifold_rows(20250211, 700, 1, d, R)[0], x[700])
    assert not np.array_equal(pyoracle.gen_manifold_rows(20250212, 0, 4, d, R), x[:4])
    nrm = pyoracle.gen_manifold_rows(20250211, 5, 3, d, R, normalize=True)
    np.testing.assert_allclose(np.linalg.norm(nrm.astype(np.float64), axis=1), 1.0, atol=1e-6)
    # the noise term is 0.05 x the EHX-GAUSS-1 element of the same (seed, row, column)
    big = pyoracle.gen_manifold_rows(20250211, 0, 2, d, 1)
    g = pyoracle.gen_rows(20250211, 0, 2, d)
    lat = pyoracle.gen_manifold_rows(20250211, 0, 2, d, 1) - np.float32(0.05) * g   # = l[r][0] * b[0][c] up to one rounding
    ratio = lat[0] / lat[1]
    assert np.nanstd(ratio.astype(np.float64)) < 1e-3 * abs(float(np.nanmean(ratio)))     # rank one


@pytest.mark.gpu
@pytest.mark.parametrize("dims,latent,normalize", [(768, 16, True), (768, 16, False), (128, 8, True), (20, 3, False),
                                                    (1536, 32, True), (64, 64, False)])
def test_device_manifold_generator_matches_oracle_bit_for_bit(dims, latent, normalize):
    import torch
    from embeddinghub_amd import _lib
    L = _lib.load()
    n, row0 = 519, 98765432101
    out = torch.empty((n, dims), dtype=torch.float32, device="cuda")
    _lib.check(L.ehx_gen_manifold_rows_device(C.c_void_p(torch.cuda.current_stream().cuda_stream), 20250211, row0, n,
                                              dims, latent, int(normalize), C.c_void_p(out.data_ptr())))
    torch.cuda.synchronize()
    ref = pyoracle.gen_manifold_rows(20250211, row0, n, dims, latent, normalize=normalize)
    assert out.cpu().numpy().tobytes() == ref.tobytes()


@pytest.mark.gpu
def test_fill_manifold_equals_set_of_oracle_rows_flat_and_graph():
    import embeddinghub_amd as ehx
    d, n, R = 96, 5000, 12
    X = pyoracle.gen_manifold_rows(ehx.SEED_CORPUS, 0, n, d, R, normalize=True)
    Q = pyoracle.gen_manifold_rows(ehx.SEED_QUERY, 0, 40, d, R, normalize=True)
    oids, odist, _ = pyoracle.exhaustive(X, Q, 10, pyoracle.METRIC_COSINE)
    a = ehx.Space.unique("man", d, metric=ehx.METRIC_COSINE)
    a.fill_manifold(ehx.SEED_CORPUS, 0, n, R, True)
    np.testing.assert_array_equal(a.get_by_id(n - 1), X[n - 1])
    ids, dist, cnt = a.knn(Q, 10)
    np.testing.assert_array_equal(ids, oids)
    assert dist.tobytes() == odist.tobytes()
    a.drop()
    g = ehx.Space.unique("man-g", d, metric=ehx.METRIC_COSINE, mode=ehx.MODE_GRAPH, initial_capacity=n)
    g.fill_manifold(ehx.SEED_CORPUS, 0, n, R, True)      # (bulk build in rounds: the graph is the engine's own)
    g.set_ef(200)
    ids, dist, cnt = g.knn(Q, 10)
    rec = np.mean([len(set(ids[i].tolist()) & set(oids[i].tolist())) / 10 for i in range(40)])
    assert rec >= 0.99, rec
    with pytest.raises(Exception):
        g.fill_manifold(ehx.SEED_CORPUS, n, 10, 65, True)
    g.drop()
