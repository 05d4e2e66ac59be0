#include "ehx_internal.h"

namespace ehx_impl {

// =====================================================================================================
// Row-sharded spaces behind the C ABI (ehx_params.shards = G > 1).
// One process drives the G devices of ehx_init's list: global row g lives in shard g % G at local row g / G; a
// search runs on every shard concurrently (one host thread and one stream per shard), each shard's local top-k
// (k * 12 + 4 bytes per query) is copied peer-to-peer over xGMI into one gather buffer on shard 0's device and
// merge_lists_kernel — the same kernel the multi-process path uses behind its RCCL all-gather — turns local rows
// into global ids (local * G + shard) and merges by (distance, id).  No other exchange step exists.
// =====================================================================================================

// f(i) for every shard, each on its own persistent thread (shard 0 on the caller's); first failure wins.  A shard that
// was dropped meanwhile (ehx_space_drop marks the parent first, so this only guards a handle that outlived its space)
// answers EHX_ENOTFOUND instead of touching released buffers.
int for_each_shard(ehx_space* p, const std::function<int(size_t)>& f) {
  if (!p->workers) return fail(EHX_EINTERNAL, "space '%s' has no shard workers", p->name.c_str());
  return p->workers->run([&](size_t i) -> int {
    if (p->shards[i]->dropped) return fail(EHX_ENOTFOUND, "Not found");
    return f(i);
  });
}

// parent locked exclusively by the caller
int sharded_set_batch(ehx_space* p, size_t n, const char* const* keys, const size_t* klens, const float* vecs) {
  if (p->frozen) return fail(EHX_EIMMUTABLE, "Cannot write to immutable space");
  if (p->implicit_keys) return fail(EHX_EINVAL, "space '%s' holds synthetic rows with implicit keys", p->name.c_str());
  const uint64_t G = p->shards.size();
  std::vector<uint64_t> ids;
  std::vector<std::string> new_keys;
  uint64_t next = 0;
  resolve_keys(p, n, keys, klens, &ids, &next, &new_keys);
  std::vector<std::vector<uint64_t>> lids(G);
  std::vector<std::vector<float>> rows(G);
  for (size_t i = 0; i < n; ++i) {
    const uint64_t sh = ids[i] % G;
    lids[sh].push_back(ids[i] / G);
    rows[sh].insert(rows[sh].end(), vecs + i * p->dims, vecs + (i + 1) * p->dims);
  }
  std::vector<uint64_t> before(G);
  for (size_t i = 0; i < G; ++i) before[i] = p->shards[i]->n;
  // Capacity first, on every shard, before any shard writes a row: the allocation that fails a batch half way (an
  // out-of-memory while one shard grows) then fails it before anything changed — graph shards cannot take rows back
  // once they are linked.
  int rc = for_each_shard(p, [&](size_t i) -> int {
    if (lids[i].empty()) return EHX_OK;
    ehx_space* c = p->shards[i];
    std::lock_guard<std::mutex> cg(c->wmu);
    std::unique_lock<std::shared_mutex> wl(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    const uint64_t next_local = (next + G - 1 - i) / G;
    int r = ensure_rows(c, next_local);
    if (!r && c->params.mode == EHX_MODE_GRAPH) r = graph_ensure_arrays(c);
    return r;
  });
  if (rc) return rc;
  rc = for_each_shard(p, [&](size_t i) -> int {
    if (lids[i].empty()) return EHX_OK;
    ehx_space* c = p->shards[i];
    std::lock_guard<std::mutex> cg(c->wmu);
    std::unique_lock<std::shared_mutex> wl(c->mu);
    const uint64_t next_local = (next + G - 1 - i) / G;  // globals below `next` that belong to shard i
    return write_rows_locked_fwd(c, lids[i].size(), lids[i], next_local, rows[i].data());
  });
  if (rc) {
    // A failing shard (e.g. out of memory while growing) must not leave the others ahead of the parent: their published
    // row counts go back to what they were, so no search returns a global id the parent has no key for.  (Rows of
    // EXISTING keys that the batch rewrote on the shards that succeeded stay rewritten — a failed batch may have
    // applied part of its updates, as a failed sequence of single Sets would; graph shards keep the nodes they linked.)
    const std::string msg = g_err;
    for (size_t i = 0; i < G; ++i) {
      ehx_space* c = p->shards[i];
      std::lock_guard<std::mutex> cg(c->wmu);
      std::unique_lock<std::shared_mutex> wl(c->mu);
      if (c->n > before[i] && c->params.mode != EHX_MODE_GRAPH) c->n = before[i];
    }
    snprintf(g_err, sizeof(g_err), "%s", msg.c_str());
    return rc;
  }
  const uint64_t old_n = p->n;
  {
    std::unique_lock<std::shared_mutex> kl(p->kmu);
    for (size_t i = 0; i < new_keys.size(); ++i) p->key_to_id.emplace(new_keys[i], old_n + i);
    for (auto& k : new_keys) p->id_to_key.push_back(std::move(k));
  }
  p->n = next;
  return EHX_OK;
}

int sharded_fill_synthetic(ehx_space* p, uint64_t seed, uint64_t row0, uint64_t n_rows, int normalize, uint32_t latent) {
  if (p->frozen) return fail(EHX_EIMMUTABLE, "Cannot write to immutable space");
  if (!p->implicit_keys && p->n != 0) return fail(EHX_EINVAL, "space '%s' already holds keyed rows", p->name.c_str());
  const uint64_t G = p->shards.size(), n0 = p->n;
  int rc = for_each_shard(p, [&](size_t i) -> int {
    // globals n0 .. n0+n_rows-1 with g % G == i: g0, g0 + G, ...; generator row of global g = row0 + (g - n0)
    const uint64_t g0 = n0 + ((i + G - n0 % G) % G);
    if (g0 >= n0 + n_rows) return EHX_OK;
    const uint64_t cnt = (n0 + n_rows - 1 - g0) / G + 1;
    ehx_space* c = p->shards[i];
    std::lock_guard<std::mutex> cg(c->wmu);
    std::unique_lock<std::shared_mutex> wl(c->mu);
    return fill_synthetic_locked(c, seed, row0 + (g0 - n0), cnt, normalize, G, latent);
  });
  if (rc) return rc;
  p->implicit_keys = true;
  p->n += n_rows;
  {
    std::unique_lock<std::shared_mutex> kl(p->kmu);
    p->implicit_n = p->n;
  }
  return EHX_OK;
}

// queries are on the host (d_queries == nullptr) or on device `qdev`; outputs likewise.  Parent locked shared.
// Per batch and shard: the queries in, the shard's own pipeline, ONE peer copy of its packed local top-k
// (ids | distances | counts: 12 k + 4 bytes per query) into its slot of the gather buffer on shard 0's device, and an
// event; the parent's stream waits for the G events (no host synchronisation per shard), merges, and hands the result
// over.  k up to 1024 like an unsharded space (every shard pages its own exhaustive pass beyond 48; the merge walks
// the lists beyond 64).
int sharded_knn(ehx_space* p, size_t nq, const float* h_queries, const float* d_queries, int qdev, uint32_t k,
                uint64_t* out_ids, float* out_dist, uint32_t* out_count, bool out_on_device, hipStream_t caller_stream) {
  if (k == 0 || nq == 0) return EHX_OK;
  if (k > 1024) return fail(EHX_EUNSUPPORTED, "k=%u exceeds 1024", k);
  const size_t G = p->shards.size();
  const int home = p->shards[0]->device;
  std::lock_guard<std::mutex> sl(p->scratch_mu);
  int rc;
  HIP_TRY(hipSetDevice(home));
  const size_t o_dist = nq * k * sizeof(uint64_t), o_cnt = o_dist + nq * k * sizeof(float);
  const size_t P = (o_cnt + nq * sizeof(uint32_t) + 15) / 16 * 16;  // one shard's packed result
  if ((rc = p->dGPack.ensure(G * P))) return rc;
  if ((rc = p->dOutIds.ensure(nq * k))) return rc;
  if ((rc = p->dOutDist.ensure(nq * k))) return rc;
  if ((rc = p->dOutCount.ensure(nq))) return rc;
  if (d_queries) {  // the caller's stream produced the queries: they must be complete before the shards read them
    HIP_TRY(hipSetDevice(qdev));
    HIP_TRY(hipStreamSynchronize(caller_stream));
  }
  const size_t qbytes = nq * p->dims * sizeof(float);
  rc = for_each_shard(p, [&](size_t i) -> int {
    ehx_space* c = p->shards[i];
    std::shared_lock<std::shared_mutex> rl(c->mu);
    std::lock_guard<std::mutex> cl(c->scratch_mu);
    HIP_TRY(hipSetDevice(c->device));
    int r;
    if ((r = c->dQraw.ensure(nq * c->dims))) return r;
    if ((r = c->dOutPack.ensure(P))) return r;
    if (!c->xev) HIP_TRY(hipEventCreateWithFlags(&c->xev, hipEventDisableTiming));
    if (d_queries) HIP_TRY(hipMemcpyPeerAsync(c->dQraw.p, c->device, d_queries, qdev, qbytes, c->stream));
    else HIP_TRY(hipMemcpyAsync(c->dQraw.p, h_queries, qbytes, hipMemcpyHostToDevice, c->stream));
    unsigned char* pk = c->dOutPack.p;
    if ((r = knn_device_locked(c, c->stream, nq, c->dQraw.p, k, (uint64_t*)pk, (float*)(pk + o_dist),
                               (uint32_t*)(pk + o_cnt))))
      return r;
    // the one exchange step
    HIP_TRY(hipMemcpyPeerAsync(p->dGPack.p + i * P, home, pk, c->device, P, c->stream));
    HIP_TRY(hipEventRecord(c->xev, c->stream));
    return EHX_OK;
  });
  if (rc) {
    // a shard failed: the others may still be writing into the gather buffer and their own scratch — drain them
    // before the error leaves (the next call reuses both); g_err keeps the failing shard's message
    for (ehx_space* c : p->shards)
      if (hipSetDevice(c->device) == hipSuccess) (void)hipStreamSynchronize(c->stream);
    (void)hipGetLastError();
    return rc;
  }
  HIP_TRY(hipSetDevice(home));
  for (size_t i = 0; i < G; ++i) HIP_TRY(hipStreamWaitEvent(p->stream, p->shards[i]->xev, 0));
  const unsigned char* gp = p->dGPack.p;
  HIP_TRY(launch_merge_lists((const uint64_t*)gp, (const float*)(gp + o_dist), (const uint32_t*)(gp + o_cnt),
                             (uint32_t)nq, k, (uint32_t)G, p->dOutIds.p, p->dOutDist.p, p->dOutCount.p, p->stream, P, P,
                             P, (uint64_t)G, 1));
  if (out_on_device) {
    HIP_TRY(hipMemcpyPeerAsync(out_ids, qdev, p->dOutIds.p, home, nq * k * sizeof(uint64_t), p->stream));
    HIP_TRY(hipMemcpyPeerAsync(out_dist, qdev, p->dOutDist.p, home, nq * k * sizeof(float), p->stream));
    HIP_TRY(hipMemcpyPeerAsync(out_count, qdev, p->dOutCount.p, home, nq * sizeof(uint32_t), p->stream));
  } else {
    HIP_TRY(hipMemcpyAsync(out_ids, p->dOutIds.p, nq * k * sizeof(uint64_t), hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipMemcpyAsync(out_dist, p->dOutDist.p, nq * k * sizeof(float), hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipMemcpyAsync(out_count, p->dOutCount.p, nq * sizeof(uint32_t), hipMemcpyDeviceToHost, p->stream));
  }
  HIP_TRY(hipStreamSynchronize(p->stream));  // (the shards' scratch may be reused by the next call: all of it is done)
  p->n_queries += nq;
  return EHX_OK;
}

}  // namespace ehx_impl
