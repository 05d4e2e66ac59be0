// Set / BatchSet (ANNIndex::set, index.cc:20-37; Go BatchSet): key resolution, pinned staging, upload, row statistics,
// scan copies, graph insertion, and the write combiner of concurrent single-row Sets.
#include "ehx_internal.h"

extern "C" {

static int set_batch_locked(ehx_space* s, size_t n, const char* const* keys, const size_t* klens, const float* vecs);
static int write_rows_locked(ehx_space* s, size_t n, const std::vector<uint64_t>& ids, uint64_t next, const float* vecs,
                             std::vector<std::string>* new_keys, bool append_only = false);
static bool all_fresh_keys(const ehx_space* s, size_t n, const std::vector<uint64_t>& ids);

int ehx_set_batch(ehx_space* s, size_t n, const char* const* keys, const size_t* klens, const float* vecs) {
  if (!valid_space(s)) return fail(EHX_EINVAL, "space is NULL");
  if (n == 0) return EHX_OK;
  if (!keys || !klens || !vecs) return fail(EHX_EINVAL, "NULL argument");
  std::lock_guard<std::mutex> wg(s->wmu);
  if (!is_parent(s) && s->params.mode == EHX_MODE_FLAT) {
    // Streaming fast path (copy.go's BatchSet chunks, MultiSet): a batch made only of fresh keys is a pure append.
    // The key lookup needs the lock shared only, and the upload runs with no lock on the space at all.
    std::vector<uint64_t> ids;
    std::vector<std::string> new_keys;
    uint64_t next = 0;
    bool fast = false;
    {
      std::shared_lock<std::shared_mutex> rl(s->mu);
      if (s->dropped) return fail(EHX_ENOTFOUND, "Not found");
      if (s->keyless) return fail(EHX_EINVAL, "a shard is written through its parent space");
      if (!s->frozen) {
        resolve_keys(s, n, keys, klens, &ids, &next, &new_keys);
        fast = new_keys.size() == n && all_fresh_keys(s, n, ids);
      }
    }
    if (fast) return write_rows_locked(s, n, ids, next, vecs, &new_keys, true);
  }
  std::unique_lock<std::shared_mutex> wl(s->mu);
  if (s->dropped) return fail(EHX_ENOTFOUND, "Not found");
  if (s->keyless) return fail(EHX_EINVAL, "a shard is written through its parent space");
  auto write = [&](size_t cnt, const char* const* ks, const size_t* kl, const float* v) -> int {
    return is_parent(s) ? sharded_set_batch(s, cnt, ks, kl, v) : set_batch_locked(s, cnt, ks, kl, v);
  };
  if (s->params.mode == EHX_MODE_GRAPH && n > 1 && s->params.build_batch != 0xFFFFFFFFu) {
    // graph mode replays a batch in call order; when it re-writes keys (known ones, or the same key
    // twice) every row must be in HBM exactly when its turn comes, so such batches go row by row
    bool rewrite = false;
    {
      std::set<std::string> seen;
      std::shared_lock<std::shared_mutex> kl(s->kmu);
      for (size_t i = 0; i < n && !rewrite; ++i) {
        std::string k(keys[i], klens[i]);
        uint64_t known = 0;
        rewrite = implicit_id(s, keys[i], klens[i], &known) || s->key_to_id.count(k) != 0 || !seen.insert(std::move(k)).second;
      }
    }
    if (rewrite) {
      for (size_t i = 0; i < n; ++i) {
        int rc = write(1, keys + i, klens + i, vecs + i * s->dims);
        if (rc) return rc;
      }
      return EHX_OK;
    }
  }
  return write(n, keys, klens, vecs);
}

// rows -> pinned staging (fp16 spaces: rounded to binary16, round-to-nearest-even, on the way); large slabs are split
// over four threads
static void stage_rows(char* dst, const float* src, size_t elems, bool half) {
  auto work = [=](size_t e0, size_t e1) {
    if (half) {
      _Float16* h = (_Float16*)dst;
      for (size_t e = e0; e < e1; ++e) h[e] = (_Float16)src[e];
    } else {
      memcpy(dst + e0 * sizeof(float), src + e0, (e1 - e0) * sizeof(float));
    }
  };
  constexpr size_t kThreads = 4;
  if (elems * sizeof(float) < (2u << 20)) {
    work(0, elems);
    return;
  }
  const size_t per = ((elems + kThreads - 1) / kThreads + 63) & ~(size_t)63;
  std::thread th[kThreads - 1];
  size_t started = 0, done_to = std::min(elems, per);  // [0, per) is this thread's share
  for (size_t t = 1; t < kThreads && t * per < elems; ++t) {
    try {
      th[t - 1] = std::thread(work, t * per, std::min(elems, (t + 1) * per));
      ++started;
      done_to = std::min(elems, (t + 1) * per);
    } catch (const std::system_error&) {
      break;  // no thread to be had (a process at its thread limit): the caller's thread copies the rest
    }
  }
  work(0, std::min(elems, per));
  for (size_t t = 0; t < started; ++t) th[t].join();
  if (done_to < elems) work(done_to, elems);
}

}  // extern "C"

namespace ehx_impl {

// wait for a stream of the space: the writers' stream through the blocking event, any other by hipStreamSynchronize
int sync_stream(ehx_space* s, hipStream_t st) {
  if (st == s->wstream && s->wev) {
    HIP_TRY(hipEventRecord(s->wev, st));
    HIP_TRY(hipEventSynchronize(s->wev));
  } else {
    HIP_TRY(hipStreamSynchronize(st));
  }
  return EHX_OK;
}

// (re)build the derived copies of rows [row0, row0+n) after they were written; must follow row_stats:
// graph mode: the search copy; flat fp32 spaces: the fp16 scan copy
// exclusive: no search can be reading the space (the caller holds s->mu exclusively and the writer's stream has waited
// for the searches in flight) — rows below the published count may then move inside their tiles; otherwise every row
// of [row0, row0 + n) lies beyond the published row count.  n_after: the row count once this write is published.
int refresh_scan16(ehx_space* s, uint64_t row0, uint64_t n, hipStream_t st, bool exclusive, uint64_t n_after) {
  if (!st) st = s->stream;
  if (s->dXs && !s->x_perm && n)
    HIP_TRY(launch_make_search_copy(s->dX, s->x_half, s->dInv, row0, n, s->ld, s->metric, s->dXs, st));
  if ((!s->has16 && !s->has8) || n == 0) return EHX_OK;  // (kept current whatever engine is selected right now)
  unsigned long long u = 0, u8[2] = {0, 0};
  if (s->has16) {
    HIP_TRY(launch_make_scan16(s->dX, s->x_half, row0, n, s->dims, s->ld, s->ld16, s->metric, s->dX16, s->dRowp16,
                               s->dUnsafe, st));
    HIP_TRY(hipMemcpyAsync(&u, s->dUnsafe, sizeof(u), hipMemcpyDeviceToHost, st));
  }
  if (s->has8) {
    // Full tiles are stored ordered by quantisation step (k_misc.hip).  Re-ordering moves rows inside a tile, so it
    // happens only where no scan can look: the fresh rows of an append (a tile that straddles the published row count
    // keeps the row order, for good), or anywhere under an exclusive writer — which re-makes whole tiles, because a
    // rewritten row of an ordered tile no longer sits where its id says.
    uint64_t r8 = row0, e8 = row0 + n;
    const bool sort_tiles = env().i8_sort;
    if (exclusive) {
      {  // searches still in flight on other streams
        int rcw = wait_searches_in_flight(s, st);
        if (rcw) return rcw;
      }
      r8 = row0 & ~(uint64_t)255;
      e8 = std::min<uint64_t>(round_up(row0 + n, 256), std::max<uint64_t>(n_after, row0 + n));
    }
    int rc8;
    const uint64_t slo = sort_tiles ? r8 : 0, shi = sort_tiles ? e8 : 0;
    if ((rc8 = s->dTileList.ensure((make_scan8_scratch_bytes(r8, e8 - r8, slo, shi) + 7) / 8))) return rc8;
    HIP_TRY(launch_make_scan8(s->dX, s->x_half, r8, e8 - r8, s->dims, s->ld, s->ld8, s->metric, s->dX8, s->dRowp8,
                              s->dTilep8, s->dPerm8, s->dTileg8, slo, shi, s->dTileList.p, s->dUnsafe8, st));
    if (s->dTileList.n > (64u << 20) / 8) {  // (a bulk load's scratch — 9 bytes per row — is not kept)
      HIP_TRY(hipStreamSynchronize(st));
      s->dTileList.release();
    }
    HIP_TRY(hipMemcpyAsync(u8, s->dUnsafe8, sizeof(u8), hipMemcpyDeviceToHost, st));
  }
  {
    int rcs = sync_stream(s, st);
    if (rcs) return rcs;
  }
  s->h_unsafe = u;
  s->h_unsafe8 = u8[0];
  if (s->has8) s->h_margin8 = u8[1];   // (cumulative: only ever grows — using the margins is sound either way)
  return EHX_OK;
}


}  // namespace ehx_impl

extern "C" {

static int set_batch_locked(ehx_space* s, size_t n, const char* const* keys, const size_t* klens, const float* vecs) {
  if (s->frozen) return fail(EHX_EIMMUTABLE, "Cannot write to immutable space");
  std::vector<uint64_t> ids;
  std::vector<std::string> new_keys;
  uint64_t next = 0;
  resolve_keys(s, n, keys, klens, &ids, &next, &new_keys);
  return write_rows_locked(s, n, ids, next, vecs, &new_keys);
}

// every key of the batch is new and distinct: the rows are a pure append
static bool all_fresh_keys(const ehx_space* s, size_t n, const std::vector<uint64_t>& ids) {
  for (size_t i = 0; i < n; ++i)
    if (ids[i] != s->n + i) return false;
  return true;
}

// rows `vecs[i]` -> row ids[i] of the space (ids < next; ids >= s->n are appended, dense), then statistics, derived
// copies, graph; finally publishes the keys (new_keys, in id order from s->n) and the new row count `next`.
//   append_only = false: the caller holds s->mu exclusively (rows may be rewritten in place, graphs change).
//   append_only = true : flat spaces, every id >= s->n.  The caller holds s->wmu only: searches keep running while
//     the rows are uploaded, described and copied BEYOND the published row count (every kernel masks rows >= n),
//     on the writers' stream; s->mu is taken exclusively just to grow the arrays (rare) and to publish.
static int write_rows_locked(ehx_space* s, size_t n, const std::vector<uint64_t>& ids, uint64_t next, const float* vecs,
                             std::vector<std::string>* new_keys, bool append_only) {
  if (s->frozen) return fail(EHX_EIMMUTABLE, "Cannot write to immutable space");
  HIP_TRY(hipSetDevice(s->device));
  hipStream_t ws = s->wstream ? s->wstream : s->stream;
  // rows rewritten in place: in-flight device searches (enqueued without the lock being held any more) finish first
  if (!append_only) {
    int rcw = wait_searches_in_flight(s, ws);
    if (rcw) return rcw;
  }
  const uint64_t old_n = s->n;
  int rc;
  if (append_only && next >= s->cap) {
    std::unique_lock<std::shared_mutex> gl(s->mu);  // the arrays move: no search may be running
    if (s->dropped) return fail(EHX_ENOTFOUND, "Not found");
    rc = ensure_rows(s, next);
  } else {
    rc = ensure_rows(s, next);
  }
  if (rc) return rc;
  // upload through pinned staging in slabs; rows may be non-contiguous (updates) so copy per row
  // (fp16 spaces: rows are rounded to binary16, round-to-nearest-even, while they are staged)
  // Two staging halves, ping-pong: slab i is copied into its half (by up to four host threads — one core moves
  // ~8 GB/s, a 25-MB chunk of copy.go's 8192 x 768 rows would spend 3 ms there) while slab i-1 is on the wire.
  const size_t row_bytes = (size_t)s->dims * s->esz;
  const size_t slab_rows = std::max<size_t>(1, std::min<size_t>(n, (8u << 20) / row_bytes));
  const size_t half_bytes = (slab_rows * row_bytes + 255) & ~(size_t)255;
  if ((rc = ensure_stage(s, 2 * half_bytes))) return rc;
  uint64_t min_id = ~0ull, max_id = 0;
  size_t slab = 0;
  // single-copy graph space: everything fallible that does not depend on the upload happens BEFORE the first row lands
  // (the id list of a non-contiguous batch and its device buffer); a failure after an in-place upload of committed rows
  // poisons the space (the rows would stay in raw order inside a permuted store)
  bool perm_run = true;
  std::vector<uint64_t> perm_uniq;
  bool touches_committed = false;
  if (s->x_perm) {
    for (size_t i = 1; i < n && perm_run; ++i) perm_run = ids[i] == ids[0] + i;
    if (!perm_run) {
      perm_uniq.assign(ids.begin(), ids.begin() + n);
      std::sort(perm_uniq.begin(), perm_uniq.end());
      perm_uniq.erase(std::unique(perm_uniq.begin(), perm_uniq.end()), perm_uniq.end());
      if ((rc = s->dPermIds.ensure(perm_uniq.size()))) return rc;
    }
    for (size_t i = 0; i < n && !touches_committed; ++i) touches_committed = ids[i] < old_n;
  }
  struct Poison {   // armed while raw rows may sit in a permuted store
    ehx_space* s;
    bool armed = false;
    ~Poison() { if (armed) s->poisoned.store(true); }
  } poison{s};
  for (size_t i0 = 0; i0 < n; i0 += slab_rows, ++slab) {
    if (touches_committed) poison.armed = true;
    const size_t m = std::min(slab_rows, n - i0);
    char* stage = (char*)s->hStage + (slab & 1) * half_bytes;
    if (slab >= 2) HIP_TRY(hipEventSynchronize(s->sev[slab & 1]));  // the upload that last used this half
    stage_rows(stage, vecs + i0 * s->dims, m * s->dims, s->x_half);
    // contiguous run of fresh ids -> one 2D copy; otherwise row by row
    bool contiguous = true;
    for (size_t i = 1; i < m; ++i)
      if (ids[i0 + i] != ids[i0] + i) { contiguous = false; break; }
    if (contiguous) {
      HIP_TRY(hipMemcpy2DAsync(s->xrow(ids[i0]), (size_t)s->ld * s->esz, stage, row_bytes,
                               row_bytes, m, hipMemcpyHostToDevice, ws));
    } else {
      for (size_t i = 0; i < m; ++i)
        HIP_TRY(hipMemcpyAsync(s->xrow(ids[i0 + i]), stage + i * row_bytes, row_bytes,
                               hipMemcpyHostToDevice, ws));
    }
    HIP_TRY(hipEventRecord(s->sev[slab & 1], ws));
    for (size_t i = 0; i < m; ++i) {
      min_id = std::min(min_id, ids[i0 + i]);
      max_id = std::max(max_id, ids[i0 + i]);
    }
  }
  // (the stream is waited for below, before the commit: both halves are free again when this call returns)
  if (s->x_perm) {
    // single-copy graph space: the rows just written go into the search copy's block order, in place, exactly once
    // each (the permutation is its own inverse: a row written twice in this batch is permuted once)
    if (perm_run) {
      HIP_TRY(launch_permute_blocks((float*)s->dX, s->ld, ids[0], n, nullptr, ws));
    } else {
      HIP_TRY(hipMemcpyAsync(s->dPermIds.p, perm_uniq.data(), perm_uniq.size() * sizeof(uint64_t), hipMemcpyHostToDevice, ws));
      HIP_TRY(launch_permute_blocks((float*)s->dX, s->ld, 0, perm_uniq.size(), s->dPermIds.p, ws));
    }
    HIP_TRY(hipStreamSynchronize(ws));  // (the list lives on this stack frame; the rows are in block order from here on)
    poison.armed = false;
  }
  // per-row statistics over the touched id range (idempotent for untouched rows in between)
  HIP_TRY(launch_row_stats(s->dX, s->x_half, min_id, max_id - min_id + 1, s->dims, s->ld, s->metric, s->dInv,
                           s->dRowp, s->dMaxSumsq, ws, s->x_perm ? 1 : 0));
  if ((rc = refresh_scan16(s, min_id, max_id - min_id + 1, ws, !append_only, next))) return rc;
  if ((rc = sync_stream(s, ws))) return rc;
  // commit: the rows are resident and described — publish the keys and the new row count
  // (the keys first, under their own lock — searches keep running — then the row count: one release store)
  if (new_keys) {
    std::shared_lock<std::shared_mutex> rl(s->mu, std::defer_lock);
    if (append_only) {
      rl.lock();  // (shared: keeps a drop out, not the searches)
      if (s->dropped) return fail(EHX_ENOTFOUND, "Not found");
    }
    std::unique_lock<std::shared_mutex> kl(s->kmu);
    for (size_t i = 0; i < new_keys->size(); ++i) s->key_to_id.emplace((*new_keys)[i], old_n + i);
    for (auto& k : *new_keys) s->id_to_key.push_back(std::move(k));
  }
  if (append_only) {
    // (shared: keeps a drop and a re-allocation out, not the searches — the count is atomic, see ehx_internal.h; the exclusive
    // lock this used to take starved behind two pipelined search callers: 63 ms per chunk at 12.5 M x 1536 under search)
    std::shared_lock<std::shared_mutex> pl(s->mu);
    if (s->dropped) return fail(EHX_ENOTFOUND, "Not found");
    s->n.store(next, std::memory_order_release);
  } else {
    s->n.store(next, std::memory_order_release);   // (the caller holds the space's lock exclusively)
  }
  if (s->params.mode == EHX_MODE_GRAPH) {
    // new rows join the graph one at a time, in id order (ANNIndex::set -> addPoint, index.cc:36);
    // rows overwritten in place keep their links (hnswlib's updatePoint repair is not built yet)
    // in call order: a fresh key is an insertion, a known key hnswlib's update-in-place
    if (s->g_n == old_n && s->params.build_batch != 0xFFFFFFFFu) {
      if ((rc = graph_ensure_arrays(s))) return rc;
      // Opt-in bulk write (ehx_params.build_batch > 1 given explicitly): a batch made only of fresh keys
      // joins the graph in concurrent rounds of up to build_batch rows — hnswlib's multi-threaded
      // add_items (offlinehub.py:89) — instead of one row per round.
      bool all_fresh = s->params.build_batch > 1 && next - old_n == n;
      for (size_t i = 0; i < n && all_fresh; ++i) all_fresh = ids[i] == old_n + i;
      if (all_fresh) return graph_insert(s, old_n, n, s->params.build_batch);
      for (size_t i = 0; i < n; ++i) {
        if (ids[i] >= s->g_n) {
          if ((rc = graph_insert(s, ids[i], 1, 1))) return rc;
        } else {
          if ((rc = graph_update(s, (uint32_t)ids[i]))) return rc;
        }
      }
    } else {
      s->g_stale_updates += n - (next - old_n);
    }
  }
  return EHX_OK;
}

}  // extern "C"

namespace ehx_impl {
int write_rows_locked_fwd(ehx_space* s, size_t n, const std::vector<uint64_t>& ids, uint64_t next, const float* vecs) {
  return write_rows_locked(s, n, ids, next, vecs, nullptr);
}
}  // namespace ehx_impl

extern "C" {

// Single-row Sets (the reference's usage: one Set per RPC / per goroutine, runner/copy.go:146-161 runs 500 at a time)
// are combined like the single-query searches are: the first caller becomes the leader, takes every request that
// queued up meanwhile (up to 4096) and writes them as ONE batch; under load the batch size grows by itself.  A call
// returns after its row is published, so a following ehx_knn from the same thread sees it (index_test.cc:39-49).
constexpr size_t kCombineMaxBatch = 4096;

int ehx_set(ehx_space* s, const char* key, size_t klen, const float* vec) {
  if (!valid_space(s)) return fail(EHX_EINVAL, "space is NULL");
  if (!key || !vec) return fail(EHX_EINVAL, "key / vector is NULL");
  ehx_space::SetReq me;
  me.key = key;
  me.klen = klen;
  me.vec = vec;
  std::unique_lock<std::mutex> lk(s->wq_mu);
  s->wq.push_back(&me);
  std::vector<ehx_space::SetReq*> group;
  std::vector<const char*> ks;
  std::vector<size_t> kl;
  std::vector<float> rows;
  while (!me.done) {
    if (s->wq_leader) {
      s->wq_cv.wait(lk, [&] { return me.done || !s->wq_leader; });
      continue;
    }
    s->wq_leader = true;
    while (!me.done && !s->wq.empty()) {
      const size_t m = std::min(s->wq.size(), kCombineMaxBatch);
      group.assign(s->wq.begin(), s->wq.begin() + m);
      s->wq.erase(s->wq.begin(), s->wq.begin() + m);
      lk.unlock();
      int rc;
      if (m == 1) {
        const char* k1[1] = {group[0]->key};
        size_t l1[1] = {group[0]->klen};
        rc = ehx_set_batch(s, 1, k1, l1, group[0]->vec);
      } else {
        ks.resize(m);
        kl.resize(m);
        rows.resize(m * s->dims);
        for (size_t i = 0; i < m; ++i) {
          ks[i] = group[i]->key;
          kl[i] = group[i]->klen;
          memcpy(rows.data() + i * s->dims, group[i]->vec, s->dims * sizeof(float));
        }
        rc = ehx_set_batch(s, m, ks.data(), kl.data(), rows.data());
        s->n_combined_batches += 1;
        s->n_combined_sets += m;
      }
      lk.lock();
      for (auto* r : group) {
        r->rc = rc;
        if (rc) snprintf(r->err, sizeof(r->err), "%s", g_err);
        r->done = true;
      }
      s->wq_cv.notify_all();
    }
    s->wq_leader = false;
    s->wq_cv.notify_all();
  }
  lk.unlock();
  if (me.rc) snprintf(g_err, sizeof(g_err), "%s", me.err);
  return me.rc;
}

}  // extern "C"
