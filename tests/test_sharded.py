"""N > 1 path on CPU: world_size-2 gloo run of the row-shard / all-gather / merge plumbing
(embeddinghub_amd/sharded.py) with the two device steps replaced by oracle-backed stand-ins, checked
against the oracle's exhaustive search over the whole index; plus the GPU merge kernel vs numpy."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from embeddinghub_amd.sharded import ShardedSearcher, shard_range
from oracle import pyoracle

N, D, B, K = 2003, 24, 37, 10


def _data():
    rng = np.random.default_rng(11)
    X = rng.standard_normal((N, D)).astype(np.float32)
    X[1500] = X[3]  # cross-shard exact tie -> (dist, id) order decides
    Q = np.concatenate([X[:5], rng.standard_normal((B - 5, D)).astype(np.float32)])
    return X, Q


def _np_merge(g_ids, g_dist, g_count, k, out_ids, out_dist, out_count):
    G, nq = g_ids.shape[0], g_ids.shape[1]
    for q in range(nq):
        items = []
        for g in range(G):
            c = int(g_count[g, q])
            items += [(float(g_dist[g, q, j]), int(g_ids[g, q, j])) for j in range(c)]
        items.sort()
        c = min(k, len(items))
        out_count[q] = c
        for j in range(c):
            out_dist[q, j], out_ids[q, j] = items[j]


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    X, Q = _data()
    row0, rows = shard_range(N, world, rank)
    Xs = X[row0:row0 + rows]

    def local_search(queries, k, ids, dst, cnt):
        i, d, c = pyoracle.exhaustive(Xs, queries.numpy(), k, pyoracle.METRIC_L2, threads=1)
        ids.copy_(torch.from_numpy(i.astype(np.int64)))
        dst.copy_(torch.from_numpy(d))
        cnt.copy_(torch.from_numpy(c.astype(np.int32)))

    s = ShardedSearcher(row0, B, K, "cpu", local_search=local_search, merge=_np_merge)
    ids, dst, cnt = s.knn(torch.from_numpy(Q))
    ret[rank] = (ids.numpy().copy(), dst.numpy().copy(), cnt.numpy().copy())
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_partitions_every_row_once():
    for n, w in [(10_000_000, 8), (2003, 2), (7, 8), (50_000_000, 8)]:
        ranges = [shard_range(n, w, r) for r in range(w)]
        assert ranges[0][0] == 0 and sum(r[1] for r in ranges) == n
        for (a0, a1), (b0, _) in zip(ranges, ranges[1:]):
            assert a0 + a1 == b0


def test_two_rank_gloo_matches_global_exhaustive():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    X, Q = _data()
    oids, odist, ocnt = pyoracle.exhaustive(X, Q, K, pyoracle.METRIC_L2)
    for rank in (0, 1):  # every rank holds the merged result
        ids, dst, cnt = ret[rank]
        np.testing.assert_array_equal(cnt, ocnt.astype(np.int32))
        np.testing.assert_array_equal(ids, oids.astype(np.int64))
        assert dst.tobytes() == odist.tobytes()
    assert 3 in ret[0][0][3] and 1500 in ret[0][0][3]  # the cross-shard tie, id order


@pytest.mark.gpu
def test_gpu_merge_kernel_matches_numpy():
    import ctypes as C
    from embeddinghub_amd import _lib
    L = _lib.load()
    rng = np.random.default_rng(3)
    G, nq, k = 8, 129, 10
    g_dist = np.sort(rng.standard_normal((G, nq, k)).astype(np.float32), axis=2)
    g_dist[:, :, 3] = g_dist[:, :, 2]  # ties inside and across lists
    g_dist[1, :, :4] = g_dist[0, :, :4]
    g_ids = rng.permutation(G * nq * k).reshape(G, nq, k).astype(np.int64)
    # lists must be (dist, id)-sorted, as ehx_knn emits them
    for g in range(G):
        for q in range(nq):
            order = np.lexsort((g_ids[g, q], g_dist[g, q]))
            g_dist[g, q], g_ids[g, q] = g_dist[g, q][order], g_ids[g, q][order]
    g_cnt = rng.integers(0, k + 1, size=(G, nq)).astype(np.int32)
    g_cnt[:, 0] = 0  # a query nobody has results for
    e_ids, e_dist, e_cnt = np.full((nq, k), -1, np.int64), np.full((nq, k), np.inf, np.float32), np.zeros(nq, np.int32)
    _np_merge(g_ids, g_dist, g_cnt, k, e_ids, e_dist, e_cnt)
    t = lambda a: torch.from_numpy(a).cuda()  # noqa: E731
    d_ids, d_dist, d_cnt = t(g_ids), t(g_dist), t(g_cnt)
    o_ids = torch.empty((nq, k), dtype=torch.int64, device="cuda")
    o_dist = torch.empty((nq, k), dtype=torch.float32, device="cuda")
    o_cnt = torch.empty((nq,), dtype=torch.int32, device="cuda")
    _lib.check(L.ehx_merge_topk_device(C.c_void_p(torch.cuda.current_stream().cuda_stream), nq, k, G,
                                       C.c_void_p(d_ids.data_ptr()), C.c_void_p(d_dist.data_ptr()),
                                       C.c_void_p(d_cnt.data_ptr()), C.c_void_p(o_ids.data_ptr()),
                                       C.c_void_p(o_dist.data_ptr()), C.c_void_p(o_cnt.data_ptr())))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(o_cnt.cpu().numpy(), e_cnt)
    for q in range(nq):
        c = e_cnt[q]
        np.testing.assert_array_equal(o_ids.cpu().numpy()[q, :c], e_ids[q, :c])
        np.testing.assert_array_equal(o_dist.cpu().numpy()[q, :c], e_dist[q, :c])


@pytest.mark.gpu
def test_gpu_packed_gather_buffer_merges_like_the_natural_layout():
    """The engine-side merge of ShardedSearcher reads ids/dist/count straight out of the packed all-gather
    buffer (list stride = packed size): same result as the natural [G][nq][k] layout."""
    from embeddinghub_amd.sharded import ShardedSearcher, _engine_merge
    rng = np.random.default_rng(4)
    G, nq, k = 4, 77, 10
    ss = ShardedSearcher(0, nq, k, "cuda", local_search=lambda *a: None, merge=lambda *a: None)
    ss.world = G  # build the gather-side views by hand (no process group in this test)
    g_pack = torch.zeros(G * ss._P, dtype=torch.uint8, device="cuda")
    g_ids, g_dst, g_cnt = ss._views(g_pack.view(G, ss._P))
    dist = np.sort(rng.standard_normal((G, nq, k)).astype(np.float32), axis=2)
    ids = np.arange(G * nq * k, dtype=np.int64).reshape(G, nq, k)
    cnt = rng.integers(0, k + 1, size=(G, nq)).astype(np.int32)
    g_ids.copy_(torch.from_numpy(ids))
    g_dst.copy_(torch.from_numpy(dist))
    g_cnt.copy_(torch.from_numpy(cnt))
    e_ids, e_dist, e_cnt = np.full((nq, k), -1, np.int64), np.full((nq, k), np.inf, np.float32), np.zeros(nq, np.int32)
    _np_merge(ids, dist, cnt, k, e_ids, e_dist, e_cnt)
    o_ids = torch.empty((nq, k), dtype=torch.int64, device="cuda")
    o_dist = torch.empty((nq, k), dtype=torch.float32, device="cuda")
    o_cnt = torch.empty((nq,), dtype=torch.int32, device="cuda")
    _engine_merge(torch.cuda.current_stream().cuda_stream)(g_ids, g_dst, g_cnt, k, o_ids, o_dist, o_cnt)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(o_cnt.cpu().numpy(), e_cnt)
    for q in range(nq):
        c = e_cnt[q]
        np.testing.assert_array_equal(o_ids.cpu().numpy()[q, :c], e_ids[q, :c])
        np.testing.assert_array_equal(o_dist.cpu().numpy()[q, :c], e_dist[q, :c])


_RCCL_ONE_RANK = r"""
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.environ["EHX_REPO"])
import embeddinghub_amd as ehx
from embeddinghub_amd.sharded import ShardedSearcher
from oracle import pyoracle
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
n, d, B, k, row0 = 30000, 96, 64, 10, 1000
X = pyoracle.gen_rows(ehx.SEED_CORPUS, row0, n, d, normalize=True)
Q = pyoracle.gen_rows(ehx.SEED_QUERY, 0, B, d, normalize=True)
s = ehx.Space.unique("rccl1", d, metric=ehx.METRIC_COSINE, initial_capacity=n)
s.fill_synthetic(ehx.SEED_CORPUS, row0, n, True)
ss = ShardedSearcher(row0, B, k, "cuda", space=s, stream=torch.cuda.current_stream().cuda_stream, exchange=True)
q = torch.from_numpy(Q).cuda()
for _ in range(3):   # (the gather buffer and the packed buffer are reused batch after batch)
    ids, dst, cnt = ss.knn(q)
torch.cuda.synchronize()
o_ids, o_dist, o_cnt = pyoracle.exhaustive(X, Q, k, pyoracle.METRIC_COSINE)
assert (cnt.cpu().numpy() == o_cnt).all()
assert (ids.cpu().numpy().astype(np.uint64) == o_ids + row0).all(), "ids differ"
assert dst.cpu().numpy().tobytes() == o_dist.tobytes(), "distance bytes differ"
dist.barrier()
dist.destroy_process_group()
print("RCCL-ONE-RANK-OK")
"""


@pytest.mark.gpu
def test_rccl_exchange_step_in_a_world_of_one_rank():
    """The one-process-per-GPU path on the hardware a test box has: ONE rank, backend "nccl" (= RCCL), the real
    all_gather_into_tensor of the packed local top-k and the engine's merge out of the gather buffer, on the streams
    bench.py uses; results = the oracle's exhaustive scan with global ids.  (Two ranks need two GPUs: RCCL refuses two
    ranks on one device; the two-rank logic runs on gloo above and in tests/test_bench_flow.py.)"""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), EHX_REPO=repo,
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    r = subprocess.run([sys.executable, "-c", _RCCL_ONE_RANK], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "RCCL-ONE-RANK-OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
