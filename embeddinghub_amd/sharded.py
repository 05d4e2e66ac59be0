"""Row-sharded kNN across the GPUs of one node: one process per GPU, `torch.distributed` (backend
"nccl" = RCCL over xGMI), ONE exchange step per batch.

Rank r holds rows [row0, row0 + rows) of the global index as its own engine space.  Every rank
searches its shard for the same query batch, local row ids are lifted to global ids, the
(ids, dist, count) triples — 12 B per result, ~124 KB per rank at B=1024, k=10 — live in ONE packed
buffer per rank, so ONE all-gather per batch moves them, and they are merged on every rank by
`ehx_merge_topk_strided_device` (k-way merge ordered by (dist, id)) straight out of the gather buffer.
There is no other collective on the data path.

The two device steps are injectable so the partition / gather / merge plumbing can be exercised on
CPU with gloo (tests/test_sharded.py); the defaults are the engine's GPU entry points and fail
loudly without a GPU.
"""
import ctypes as C


def shard_range(n_total, world, rank):
    """Contiguous row range of `rank`: (row0, rows); the last rank takes the remainder."""
    base = n_total // world
    row0 = rank * base
    rows = base if rank < world - 1 else n_total - row0
    return row0, rows


def _engine_local_search(space, stream):
    def search(queries, k, ids, dist, count):
        space.knn_device(queries, k, ids, dist, count, stream=stream)
    return search


def _engine_merge(stream):
    from . import _lib
    L = _lib.load()

    def merge(g_ids, g_dist, g_count, k, out_ids, out_dist, out_count):
        # g_*: [n_lists, nq, k] / [n_lists, nq] views whose list stride may be that of the packed gather buffer
        n_lists, nq = g_ids.shape[0], g_ids.shape[1]
        _lib.check(L.ehx_merge_topk_strided_device(
            C.c_void_p(stream or 0), nq, k, n_lists,
            C.c_void_p(g_ids.data_ptr()), g_ids.stride(0) * 8, C.c_void_p(g_dist.data_ptr()), g_dist.stride(0) * 4,
            C.c_void_p(g_count.data_ptr()), g_count.stride(0) * 4,
            C.c_void_p(out_ids.data_ptr()), C.c_void_p(out_dist.data_ptr()), C.c_void_p(out_count.data_ptr())))
    return merge


class ShardedSearcher:
    def __init__(self, row0, batch, k, device, group=None, local_search=None, merge=None, space=None,
                 stream=None, exchange=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # exchange=True: gather + merge even in a world of one rank (how a 1-GPU box exercises the RCCL step)
        self.exchange = self.world > 1 if exchange is None else bool(exchange)
        self.row0, self.k, self.batch = int(row0), int(k), int(batch)
        self.local_search = local_search or _engine_local_search(space, stream)
        self.merge = merge or (_engine_merge(stream) if self.exchange else None)
        mk = lambda shape, dt: torch.empty(shape, dtype=dt, device=device)  # noqa: E731
        # one packed buffer per rank: ids [B,k] i64 | dist [B,k] f32 | count [B] i32, padded to 16 bytes
        B = self.batch
        self._o_dist, self._o_cnt = B * k * 8, B * k * 12
        self._P = (B * k * 12 + B * 4 + 15) // 16 * 16
        self.pack = torch.zeros(self._P, dtype=torch.uint8, device=device)
        self.ids, self.dst, self.cnt = self._views(self.pack.view(1, self._P))
        self.ids, self.dst, self.cnt = self.ids[0], self.dst[0], self.cnt[0]
        if self.exchange:
            G = self.world
            self.g_pack = torch.zeros(G * self._P, dtype=torch.uint8, device=device)
            self.g_ids, self.g_dst, self.g_cnt = self._views(self.g_pack.view(G, self._P))
            self.m_ids, self.m_dst, self.m_cnt = (mk((batch, k), torch.int64), mk((batch, k), torch.float32),
                                                  mk((batch,), torch.int32))

    def _views(self, buf):
        """[n, P] uint8 -> (ids [n,B,k] i64, dist [n,B,k] f32, count [n,B] i32) views into it"""
        t, B, k = self.torch, self.batch, self.k
        ids = buf[:, :self._o_dist].view(t.int64).unflatten(1, (B, k))
        dst = buf[:, self._o_dist:self._o_cnt].view(t.float32).unflatten(1, (B, k))
        cnt = buf[:, self._o_cnt:self._o_cnt + B * 4].view(t.int32)
        return ids, dst, cnt

    def knn(self, queries):
        """queries: [batch, dims] on `device`.  Returns (ids, dist, count) tensors (global ids)."""
        self.local_search(queries, self.k, self.ids, self.dst, self.cnt)
        if not self.exchange:
            if self.row0:
                self.ids.add_(self.row0)
            return self.ids, self.dst, self.cnt
        self.ids.add_(self.row0)  # local row id -> global id (entries beyond count are ignored by the merge)
        self.dist.all_gather_into_tensor(self.g_pack, self.pack, group=self.group)  # the one exchange step
        self.merge(self.g_ids, self.g_dst, self.g_cnt, self.k, self.m_ids, self.m_dst, self.m_cnt)
        return self.m_ids, self.m_dst, self.m_cnt
