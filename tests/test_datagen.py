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
