"""GPU tests of the exact engine on STRUCTURED rows (a low-dimensional manifold in 768 dimensions — what real embeddings
look like) and of the int8 scan's lock-step by tile (round 5), through the C ABI.

Round 4 reported the exact engine "8 x bimodal" on such rows (0.12-0.14 M queries/s in three default bench runs, 1.05-1.10 M
in two).  Round 5 found: the scan phase takes the same 0.82-0.85 ms per batch in both "modes", no query
falls back and nothing adapts; the slow runs hold ONE stall of ~60 ms inside a kernel launch of the HIP runtime among the
ten timed batches.  What the engine owes its users is pinned here: on structured rows the scan is as fast as on Gaussian
rows of the same shape, no query leaves the int8 stage once the list has adapted, and the answers are the oracle's."""
import json
import os
import subprocess
import sys
import time

import numpy as np
import pytest

from oracle import pyoracle

pytestmark = pytest.mark.gpu

ehx = pytest.importorskip("embeddinghub_amd")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def manifold(seed, rows, d, R, A):
    r = np.random.default_rng(seed)
    x = r.standard_normal((rows, R)).astype(np.float32) @ A
    x += 0.05 * r.standard_normal((rows, d)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    return np.ascontiguousarray(x, dtype=np.float32)


def gaussian(seed, rows, d):
    x = np.random.default_rng(seed).standard_normal((rows, d)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    return x


def _steady_state(space, Qs, k):
    """scan-phase ms per batch (HIP events) and wall ms per batch (median) once no batch loses queries any more"""
    import torch
    B = Qs[0].shape[0]
    dq = [torch.from_numpy(q).cuda() for q in Qs]
    ids = torch.empty((B, k), dtype=torch.int64, device="cuda")
    dst = torch.empty((B, k), dtype=torch.float32, device="cuda")
    cnt = torch.empty((B,), dtype=torch.int32, device="cuda")
    st = torch.cuda.Stream()
    for _ in range(8):   # untimed: scratch allocation; a list that has to grow does so here
        space.stats_reset()
        for q in dq:
            space.knn_device(q, k, ids, dst, cnt, stream=st.cuda_stream)
        st.synchronize()
        s = space.stats()
        if s["n_i8_fallback"] + s["n_filter_fallback"] + s["n_exhaustive"] == 0:
            break
    space.stats_reset()
    wall = []
    for rep in range(6):
        for q in dq:
            t0 = time.perf_counter()
            space.knn_device(q, k, ids, dst, cnt, stream=st.cuda_stream)
            st.synchronize()
            wall.append((time.perf_counter() - t0) * 1e3)
    s = space.stats()
    return s, float(np.median(wall))


def test_structured_rows_scan_as_fast_as_gaussian_rows_and_stay_in_the_int8_stage():
    n, d, R, B, k = 200_000, 768, 16, 1024, 10
    A = np.random.default_rng(20250213).standard_normal((R, d)).astype(np.float32) / np.sqrt(R)
    res = {}
    for name, rows, queries in (
            ("structured", lambda i, m: manifold(100 + i, m, d, R, A), lambda i: manifold(5000 + i, B, d, R, A)),
            ("gaussian", lambda i, m: gaussian(200 + i, m, d), lambda i: gaussian(6000 + i, B, d))):
        sp = ehx.Space.unique("srows-" + name, d, metric=ehx.METRIC_COSINE, initial_capacity=n)
        X = []
        for c, i0 in enumerate(range(0, n, 50_000)):
            x = rows(c, 50_000)
            X.append(x)
            sp.set_batch(["k%d" % i for i in range(i0, i0 + 50_000)], x)
        Qs = [queries(i) for i in range(3)]
        assert sp.scan_engine() == "i8"
        s, wall_ms = _steady_state(sp, Qs, k)
        assert s["n_i8_fallback"] + s["n_filter_fallback"] + s["n_exhaustive"] == 0, (name, s)
        assert s["n_i8_queries"] == 18 * B
        res[name] = (s["scan_ms_mean"], wall_ms)
        if name == "structured":   # ... and the answers are the oracle's, ids and distance bytes (64 queries)
            ids, dist, cnt = sp.knn(Qs[0][:64], k)
            oids, odist, _ = pyoracle.exhaustive(np.concatenate(X), Qs[0][:64], k, pyoracle.METRIC_COSINE)
            assert np.array_equal(ids, oids) and dist.tobytes() == odist.tobytes()
        sp.drop()
    (s_scan, s_wall), (g_scan, g_wall) = res["structured"], res["gaussian"]
    assert s_scan <= 1.3 * g_scan, "scan phase on structured rows %.3f ms against %.3f ms on Gaussian rows" % (s_scan, g_scan)
    assert s_wall <= 1.3 * g_wall + 0.1, "wall per batch (median) %.3f ms against %.3f ms" % (s_wall, g_wall)


LOCKSTEP_CHILD = r"""
import json, sys
import numpy as np
sys.path.insert(0, %(root)r)
import embeddinghub_amd as ehx
from oracle import pyoracle
n, d, nq, k = 150000, 768, 1024, 10
X = pyoracle.gen_rows(ehx.SEED_CORPUS, 0, n, d, normalize=True)
Q = pyoracle.gen_rows(ehx.SEED_QUERY, 0, nq, d, normalize=True)
s = ehx.Space.unique("lockstep", d, metric=ehx.METRIC_COSINE, initial_capacity=n)
s.fill_synthetic(ehx.SEED_CORPUS, 0, n, True)
assert s.scan_engine() == "i8"
out = []
for rep in range(3):
    ids, dist, cnt = s.knn(Q, k)
    oids, odist, _ = pyoracle.exhaustive(X, Q, k, pyoracle.METRIC_COSINE, threads=8)
    out.append(bool(np.array_equal(ids, oids) and dist.tobytes() == odist.tobytes() and (cnt == k).all()))
st = s.stats()
print(json.dumps({"identical": out, "i8_queries": int(st["n_i8_queries"]), "fallbacks": int(st["n_i8_fallback"])}))
"""


@pytest.mark.parametrize("mode", ["1", "2", "rev"])
def test_lockstep_of_sibling_workgroups_leaves_the_answers_alone(mode):
    """EHX_I8_SYNC (read once per process, hence a child): the four query-tile workgroups that stream one row chunk keep
    within N tiles of each other (by tile: round 5; "rev": by ring revolution, rounds 2-4) — a scheduling constraint only:
    ids and distance bytes stay the oracle's on a shape whose passes take the lock-step path (4 query tiles, chunks on
    whole XCDs, >= 4 tiles per chunk)"""
    env = dict(os.environ, EHX_I8_SYNC=mode)
    r = subprocess.run([sys.executable, "-c", LOCKSTEP_CHILD % {"root": ROOT}], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    got = json.loads(r.stdout.strip().splitlines()[-1])
    assert got["identical"] == [True, True, True] and got["i8_queries"] == 3 * 1024 and got["fallbacks"] == 0
