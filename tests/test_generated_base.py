"""Keyed rows after a generated base (round 6): ehx_fill_synthetic / ehx_fill_manifold give rows [0, n0) the implicit keys
"0" .. "n0-1"; rows written afterwards through ehx_set / ehx_set_batch carry their own keys, and a Set of an implicit key
rewrites that row (upsert, index.cc:21-35) — what lets bench.py stream Sets into a 12.5 M-row fp16 shard (BASELINE
configs[4]) whose base never existed on the host.  Results against the oracle's exhaustive scan of the same rows."""
import numpy as np
import pytest

from oracle import pyoracle

pytestmark = pytest.mark.gpu
ehx = pytest.importorskip("embeddinghub_amd")


@pytest.mark.parametrize("dtype", [ehx.DTYPE_F32, ehx.DTYPE_F16])
def test_keyed_rows_after_a_generated_base(dtype):
    d, n0, m = 64, 20000, 700
    s = ehx.Space.unique("genbase", d, metric=ehx.METRIC_COSINE, dtype=dtype)
    s.fill_synthetic(ehx.SEED_CORPUS, 0, n0, True)
    X = pyoracle.gen_rows(ehx.SEED_CORPUS, 0, n0, d, normalize=True)
    rng = np.random.default_rng(4)
    Y = rng.standard_normal((m, d)).astype(np.float32)
    s.set_batch(["new%d" % i for i in range(m)], Y)                  # fresh keys: appended after the base
    assert len(s) == n0 + m
    s.set("17", Y[3])                                                # an implicit key: row 17 is rewritten
    s.set_batch(["new5", "40", "brand"], Y[10:13])                   # a keyed row, an implicit row, a fresh key
    assert len(s) == n0 + m + 1
    allrows = np.concatenate([X, Y, Y[12:13]])
    allrows[17] = Y[3]
    allrows[n0 + 5] = Y[10]
    allrows[40] = Y[11]
    if dtype == ehx.DTYPE_F16:
        allrows = allrows.astype(np.float16).astype(np.float32)
    assert s.key_of(17) == "17" and s.key_of(n0 + 2) == "new2" and s.key_of(n0 + m) == "brand"
    np.testing.assert_array_equal(s.get("17"), allrows[17])
    np.testing.assert_array_equal(s.get("brand"), allrows[n0 + m])
    np.testing.assert_array_equal(s.get("123"), allrows[123])
    with pytest.raises(Exception):
        s.get("0123")                                                # not the canonical decimal: no such key
    Q = np.concatenate([pyoracle.gen_rows(ehx.SEED_QUERY, 0, 24, d, normalize=True), Y[:8]])
    ids, dist, cnt = s.knn(Q, 10)
    oids, odist, _ = pyoracle.exhaustive(allrows, Q, 10, pyoracle.METRIC_COSINE)
    np.testing.assert_array_equal(ids, oids)
    assert dist.tobytes() == odist.tobytes()
    keys = s.knn_keys(Q[24:25], 3)[0]                                # a query equal to a keyed row finds its key first
    assert keys[0] == "new0"
    with pytest.raises(Exception):
        s.fill_synthetic(ehx.SEED_CORPUS, n0, 10, True)              # generated rows only at the head of a space
    s.drop()
