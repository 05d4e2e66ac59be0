// Search entry points: ehx_knn_device, ehx_knn from host pointers (slots, pipelined int8 stage, micro-batcher), key
// lookups, top-k merge of shard lists.
#include "ehx_internal.h"

extern "C" {

int ehx_knn_device(ehx_space* s, void* stream, size_t n_queries, const float* d_queries, uint32_t k,
                   uint64_t* d_out_ids, float* d_out_dist, uint32_t* d_out_count) {
  if (!valid_space(s)) return fail(EHX_EINVAL, "space is NULL");
  if (n_queries && k && (!d_queries || !d_out_ids || !d_out_dist || !d_out_count))
    return fail(EHX_EINVAL, "NULL device pointer");
  std::shared_lock<std::shared_mutex> rl(s->mu);
  if (s->dropped) return fail(EHX_ENOTFOUND, "Not found");
  if (is_parent(s)) {
    if (n_queries == 0 || k == 0) return EHX_OK;
    hipPointerAttribute_t at;
    HIP_TRY(hipPointerGetAttributes(&at, d_queries));
    return sharded_knn(s, n_queries, nullptr, d_queries, at.device, k, d_out_ids, d_out_dist, d_out_count, true,
                       (hipStream_t)stream);
  }
  std::lock_guard<std::mutex> sl(s->scratch_mu);
  HIP_TRY(hipSetDevice(s->device));
  return knn_device_locked(s, (hipStream_t)stream, n_queries, d_queries, k, d_out_ids, d_out_dist, d_out_count);
}

static int knn_host_direct(ehx_space* s, size_t n_queries, const float* queries, uint32_t k, uint64_t* out_ids,
                           float* out_dist, uint32_t* out_count) {
  if (!valid_space(s)) return fail(EHX_EINVAL, "space is NULL");
  if (n_queries == 0) return EHX_OK;
  if (!out_count) return fail(EHX_EINVAL, "out_count is NULL");
  if (k == 0) {
    for (size_t i = 0; i < n_queries; ++i) out_count[i] = 0;
    return EHX_OK;
  }
  if (!queries || !out_ids || !out_dist) return fail(EHX_EINVAL, "NULL argument");
  std::shared_lock<std::shared_mutex> rl(s->mu);
  if (s->dropped) return fail(EHX_ENOTFOUND, "Not found");
  if (is_parent(s)) return sharded_knn(s, n_queries, queries, nullptr, 0, k, out_ids, out_dist, out_count, false, nullptr);
  HIP_TRY(hipSetDevice(s->device));
  int rc;
  const size_t qbytes = n_queries * s->dims * sizeof(float);
  // A small call (the reference's request shape: one query, ten keys) is all fixed cost: its queries go through a
  // pinned staging buffer (an asynchronous copy instead of the runtime's pageable-memory path) and its three result
  // arrays come back as ONE block into pinned memory instead of three blocking copies.
  constexpr size_t kSmallCall = 32u << 10;
  const size_t nk = n_queries * k;
  const size_t out_bytes = nk * (sizeof(uint64_t) + sizeof(float)) + n_queries * sizeof(uint32_t);
  if (qbytes <= kSmallCall && out_bytes <= kSmallCall) {
    std::lock_guard<std::mutex> sl(s->scratch_mu);
    // ONE query against a small flat shard — the reference's request (server.cc:172-210; BASELINE configs[0]): a single
    // launch reads the query from host-visible memory, scans every row in the oracle's arithmetic, and the last
    // workgroup writes the answer into host-visible memory and raises a flag this thread spins on (k_flat.hip:
    // single_query_kernel).  10 k x 128: ~130 us through the three-launch path -> see DESIGN §e.
    const uint64_t one_bytes = env().small_exact_bytes;
    const bool one_on = env().one_launch;
    if (one_on && n_queries == 1 && k <= 64 && s->params.mode == EHX_MODE_FLAT && s->scan_sel == EHX_SCAN_AUTO && s->n > 0 &&
        s->ld <= 4096 && (uint64_t)s->n * s->ld * s->esz <= one_bytes) {
      constexpr size_t kOneQ = 16384;   // query slot (ld <= 4096 floats)
      if (!s->hOnePin) {
        HIP_TRY(hipHostMalloc((void**)&s->hOnePin, kOneQ + 2048, hipHostMallocCoherent | hipHostMallocMapped));
        memset(s->hOnePin, 0, kOneQ + 2048);
      }
      if (!s->dOneTicket) {
        HIP_TRY(hipMalloc((void**)&s->dOneTicket, sizeof(uint32_t)));
        HIP_TRY(hipMemset(s->dOneTicket, 0, sizeof(uint32_t)));
      }
      const uint32_t rpb = (uint32_t)std::max<uint64_t>(64, ((s->n + 1023) / 1024 + 63) / 64 * 64);  // <= 1024 workgroups
      const uint32_t n_blocks = (uint32_t)((s->n + rpb - 1) / rpb);
      if ((rc = s->dOnePart.ensure((size_t)n_blocks * 64))) return rc;
      if ((rc = wait_searches_in_flight(s, s->stream))) return rc;  // (device searches queued on other streams)
      char* h = s->hOnePin;
      memcpy(h, queries, qbytes);
      SingleQueryArgs a;
      a.q_in = (const float*)h;
      a.X = s->dX;
      a.inv_norm = s->dInv;
      a.part = s->dOnePart.p;
      a.ticket = s->dOneTicket;
      a.out_ids = (uint64_t*)(h + kOneQ);
      a.out_dist = (float*)(h + kOneQ + 512);
      a.out_count = (uint32_t*)(h + kOneQ + 768);
      a.done_flag = (uint32_t*)(h + kOneQ + 1024);
      a.seq = ++s->one_seq ? s->one_seq : ++s->one_seq;   // (never 0: the buffer starts zeroed)
      a.x_half = (uint32_t)s->x_half;
      a.n = (uint32_t)s->n;
      a.dims = s->dims;
      a.ld = s->ld;
      a.rows_per_block = rpb;
      a.k = k;
      a.metric = s->metric;
      HIP_TRY(launch_single_query(a, n_blocks, s->stream));
      volatile uint32_t* flag = (volatile uint32_t*)a.done_flag;
      bool seen = false;
      for (uint32_t spin = 0; spin < 4000000u; ++spin) {   // ~ tens of milliseconds at most, then ask the runtime
        if (*flag == a.seq) {
          seen = true;
          break;
        }
        __builtin_ia32_pause();
      }
      if (!seen) {
        HIP_TRY(hipStreamSynchronize(s->stream));
        if (*flag != a.seq) return fail(EHX_EINTERNAL, "single-query kernel finished without publishing its result");
      }
      std::atomic_thread_fence(std::memory_order_acquire);
      memcpy(out_ids, a.out_ids, k * sizeof(uint64_t));
      memcpy(out_dist, a.out_dist, k * sizeof(float));
      out_count[0] = *a.out_count;
      s->n_queries += 1;
      s->n_exhaustive += 1;
      s->n_one_launch += 1;
      s->n_dist += s->n;
      return EHX_OK;
    }
    // ... and in GRAPH mode (round 5; the reference's own index and request: HNSW, one query per NearestNeighbor RPC):
    // the graph search kernel reads the raw query from host-visible memory, prepares it itself, walks the graph (one
    // wave) and writes the answer into host-visible memory — one launch instead of prepare + search + two copies.
    if (one_on && n_queries == 1 && k <= 64 && s->params.mode == EHX_MODE_GRAPH && s->n > 0 && s->ld <= 4096) {
      constexpr size_t kOneQ = 16384;
      if (!s->hOnePin) {
        HIP_TRY(hipHostMalloc((void**)&s->hOnePin, kOneQ + 2048, hipHostMallocCoherent | hipHostMallocMapped));
        memset(s->hOnePin, 0, kOneQ + 2048);
      }
      char* h = s->hOnePin;
      memcpy(h, queries, qbytes);
      GraphOneLaunch one;
      one.q_host = (const float*)h;
      one.done_flag = (uint32_t*)(h + kOneQ + 1024);
      one.seq = ++s->one_seq ? s->one_seq : ++s->one_seq;   // (never 0: the buffer starts zeroed)
      uint64_t* o_ids = (uint64_t*)(h + kOneQ);
      float* o_dist = (float*)(h + kOneQ + 512);
      uint32_t* o_cnt = (uint32_t*)(h + kOneQ + 768);
      if ((rc = knn_graph_locked(s, s->stream, 1, nullptr, k, o_ids, o_dist, o_cnt, &one))) return rc;
      volatile uint32_t* flag = (volatile uint32_t*)one.done_flag;
      bool seen = false;
      for (uint32_t spin = 0; spin < 4000000u; ++spin) {
        if (*flag == one.seq) {
          seen = true;
          break;
        }
        __builtin_ia32_pause();
      }
      if (!seen) {
        HIP_TRY(hipStreamSynchronize(s->stream));
        if (*flag != one.seq) return fail(EHX_EINTERNAL, "graph search kernel finished without publishing its result");
      }
      std::atomic_thread_fence(std::memory_order_acquire);
      memcpy(out_ids, o_ids, k * sizeof(uint64_t));
      memcpy(out_dist, o_dist, k * sizeof(float));
      out_count[0] = *o_cnt;
      s->n_one_launch += 1;
      return EHX_OK;
    }
    if ((rc = s->dQraw.ensure(n_queries * s->dims))) return rc;
    if (!s->hSmallPin) HIP_TRY(hipHostMalloc((void**)&s->hSmallPin, 2 * kSmallCall, hipHostMallocDefault));
    if ((rc = s->dSmallOut.ensure(kSmallCall / sizeof(uint64_t)))) return rc;
    uint64_t* d_ids = s->dSmallOut.p;
    float* d_dist = (float*)(d_ids + nk);
    uint32_t* d_cnt = (uint32_t*)(d_dist + nk);
    memcpy(s->hSmallPin, queries, qbytes);
    HIP_TRY(hipMemcpyAsync(s->dQraw.p, s->hSmallPin, qbytes, hipMemcpyHostToDevice, s->stream));
    if ((rc = knn_device_locked(s, s->stream, n_queries, s->dQraw.p, k, d_ids, d_dist, d_cnt))) return rc;
    char* h = s->hSmallPin + kSmallCall;
    HIP_TRY(hipMemcpyAsync(h, d_ids, out_bytes, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    memcpy(out_ids, h, nk * sizeof(uint64_t));
    memcpy(out_dist, h + nk * sizeof(uint64_t), nk * sizeof(float));
    memcpy(out_count, h + nk * (sizeof(uint64_t) + sizeof(float)), n_queries * sizeof(uint32_t));
    return EHX_OK;
  }
  // A batch: through a slot of its own (see ehx_space::HostSlot) — only the device pipeline itself is serialised (the
  // pipeline's lock is NOT held while the queries are staged: the first version took it on entry and two callers ran
  // strictly one after the other).
  ehx_space::HostSlot* hs = nullptr;
  {
    std::unique_lock<std::mutex> hl(s->hs_mu);
    s->hs_cv.wait(hl, [&] {
      for (auto& h : s->hslot)
        if (!h.busy) return true;
      return false;
    });
    for (auto& h : s->hslot)
      if (!h.busy) {
        hs = &h;
        break;
      }
    hs->busy = true;
  }
  struct Release {
    ehx_space* s;
    ehx_space::HostSlot* h;
    bool ok = false;   // set on the success path; an early error return may leave copies / kernels of this call in flight
    ~Release() {
      if (!ok) {  // drain them before the slot's pinned and device buffers go to the next caller
        if (h->st) (void)hipStreamSynchronize(h->st);
        if (s->stream) (void)hipStreamSynchronize(s->stream);
      }
      {
        std::lock_guard<std::mutex> hl(s->hs_mu);
        h->busy = false;
      }
      s->hs_cv.notify_one();
    }
  } release{s, hs};
  const size_t ids_b = nk * sizeof(uint64_t), dist_b = nk * sizeof(float);
  const size_t need = qbytes + out_bytes;
  if (!hs->st) {
    HIP_TRY(hipStreamCreateWithFlags(&hs->st, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&hs->in_ev, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&hs->done_ev, hipEventDisableTiming));
  }
  if (hs->pin_bytes < need) {
    HIP_TRY(hipStreamSynchronize(hs->st));
    if (hs->pin) (void)hipHostFree(hs->pin);
    hs->pin = nullptr;
    hs->pin_bytes = 0;
    HIP_TRY(hipHostMalloc((void**)&hs->pin, need, hipHostMallocDefault));
    hs->pin_bytes = need;
  }
  if ((rc = hs->dq.ensure(n_queries * s->dims))) return rc;
  if ((rc = hs->dout.ensure(out_bytes))) return rc;
  uint64_t* d_ids = (uint64_t*)hs->dout.p;
  float* d_dist = (float*)(hs->dout.p + ids_b);
  uint32_t* d_cnt = (uint32_t*)(hs->dout.p + ids_b + dist_b);
  memcpy(hs->pin, queries, qbytes);
  HIP_TRY(hipMemcpyAsync(hs->dq.p, hs->pin, qbytes, hipMemcpyHostToDevice, hs->st));
  HIP_TRY(hipEventRecord(hs->in_ev, hs->st));
  // The int8 engine's first stage — all of a batch unless queries lose their certificate — runs in one of the space's two
  // scratch sets WITHOUT the pipeline-wide lock: this call's launches queue up on the space's stream behind the other
  // caller's while that one still waits for its verdict, so the scan kernels of consecutive batches run back to back with
  // no host round trip (launches, verdict copy, thread wake-up: ~0.1 ms per batch) between them.  A batch that does lose
  // queries is re-run through the full engine chain under the lock (rare; the chain also adapts the list's length).
  const bool pipe_on = env().host_pipeline;
  bool done = false, have_failed = false, copied_early = false;
  std::vector<uint32_t> failed;
  size_t n_short = 0;
  uint32_t kprime_used = 0;
  if (pipe_on && s->params.mode == EHX_MODE_FLAT && k <= EHX_MAX_K && s->n > 0 && resolve_engine(s) == EHX_ENGINE_I8) {
    const int set = (int)(s->i8_next_set.fetch_add(1, std::memory_order_relaxed) & 1u);   // consecutive batches alternate
    ehx_space::I8Set& sc = s->i8set[set];
    std::lock_guard<std::mutex> l(sc.mu);
    std::unique_lock<std::mutex> ql(s->i8_enqueue_mu);   // (this batch's launches go onto the stream as one block)
    HIP_TRY(hipStreamWaitEvent(s->stream, hs->in_ev, 0));
    if ((rc = flat_pass8(s, set, s->stream, n_queries, hs->dq.p, k, d_ids, d_dist, d_cnt, false, &kprime_used))) return rc;
    HIP_TRY(hipMemcpyAsync(sc.hUncertPin, sc.dUncert, sizeof(unsigned long long), hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipEventRecord(sc.verdict, s->stream));
    ql.unlock();
    // the results start their way back NOW, before the verdict is known (it is clean for all but a few batches in a
    // thousand): one host wait per batch instead of two in a row — verdict, then copy.  A batch that did lose queries runs
    // the rest of the chain below and copies again (same slot stream: in order, the later copy wins).
    // (it waits for the verdict's event — recorded right behind the re-rank and the 8-byte verdict copy; a marker of its own
    // cost the queue another ~6 us per batch)
    HIP_TRY(hipStreamWaitEvent(hs->st, sc.verdict, 0));
    HIP_TRY(hipMemcpyAsync(hs->pin + qbytes, hs->dout.p, out_bytes, hipMemcpyDeviceToHost, hs->st));
    copied_early = true;
    HIP_TRY(hipEventSynchronize(sc.verdict));
    s->n_queries += n_queries;
    s->n_dist += (uint64_t)n_queries * s->n;
    s->bytes_algo += s->n * (uint64_t)s->dims + (uint64_t)n_queries * s->dims * 4ull + (uint64_t)n_queries * k * 12ull;
    if (*sc.hUncertPin == 0) {
      done = true;
      s->n_i8_queries += n_queries;
      i8_adapt(s, n_queries, 0, 0, kprime_used);   // a clean batch: the score decays
    } else {  // which queries, and why: the engine chain continues with them (below, under the pipeline lock)
      HIP_TRY(hipMemsetAsync(sc.dUncert, 0, sizeof(unsigned long long), s->stream));
      std::vector<uint32_t> flags(n_queries);
      HIP_TRY(hipMemcpyAsync(flags.data(), sc.dUflags.p, n_queries * sizeof(uint32_t), hipMemcpyDeviceToHost, s->stream));
      HIP_TRY(hipStreamSynchronize(s->stream));
      for (size_t j = 0; j < n_queries; ++j)
        if (flags[j]) {
          failed.push_back((uint32_t)j);
          n_short += flags[j] == 2u;
        }
      have_failed = true;
    }
  }
  if (!done) {
    std::lock_guard<std::mutex> sl2(s->scratch_mu);
    HIP_TRY(hipStreamWaitEvent(s->stream, hs->in_ev, 0));
    if ((rc = knn_device_locked(s, s->stream, n_queries, hs->dq.p, k, d_ids, d_dist, d_cnt, have_failed ? &failed : nullptr,
                                n_short, kprime_used)))
      return rc;
    HIP_TRY(hipEventRecord(hs->done_ev, s->stream));
  }
  char* ho = hs->pin + qbytes;
  if (!(done && copied_early)) {
    HIP_TRY(hipStreamWaitEvent(hs->st, hs->done_ev, 0));
    HIP_TRY(hipMemcpyAsync(ho, hs->dout.p, out_bytes, hipMemcpyDeviceToHost, hs->st));
  }
  HIP_TRY(hipStreamSynchronize(hs->st));
  memcpy(out_ids, ho, ids_b);
  memcpy(out_dist, ho + ids_b, dist_b);
  memcpy(out_count, ho + ids_b + dist_b, n_queries * sizeof(uint32_t));
  release.ok = true;
  return EHX_OK;
}

// Small calls (the reference's usage: one query per RPC, server.cc:172-210; Go Nearest, online.go:63) are
// coalesced: the first caller becomes the leader, gathers every request that queued up meanwhile
// (same k, up to 1024 queries), runs ONE device batch and hands the results back.  An uncontended call
// runs immediately; under load the batch size grows by itself with the scan time.
constexpr size_t kCoalesceMaxCall = 64;     // calls above this size already are batches
constexpr size_t kCoalesceMaxBatch = 1024;

int ehx_knn(ehx_space* s, size_t n_queries, const float* queries, uint32_t k, uint64_t* out_ids,
            float* out_dist, uint32_t* out_count) {
  if (!valid_space(s)) return fail(EHX_EINVAL, "space is NULL");
  if (n_queries == 0) return EHX_OK;
  if (n_queries > kCoalesceMaxCall || k == 0 || !queries || !out_ids || !out_dist || !out_count)
    return knn_host_direct(s, n_queries, queries, k, out_ids, out_dist, out_count);
  ehx_space::KnnReq me;
  me.q = queries;
  me.nq = n_queries;
  me.k = k;
  me.ids = out_ids;
  me.dist = out_dist;
  me.cnt = out_count;
  std::unique_lock<std::mutex> lk(s->bq_mu);
  s->bq.push_back(&me);
  std::vector<ehx_space::KnnReq*> group;
  std::vector<float> q;
  std::vector<uint64_t> ids;
  std::vector<float> dist;
  std::vector<uint32_t> cnt;
  while (!me.done) {
    if (s->bq_leader) {  // somebody else is serving: wait for my result, or for the leadership to come free
      s->bq_cv.wait(lk, [&] { return me.done || !s->bq_leader; });
      continue;
    }
    // Leader: serve groups until my own request has been answered, then hand the role to a waiter (a leader
    // that kept serving while the queue refills would delay its own, already answered, caller without bound).
    s->bq_leader = true;
    while (!me.done && !s->bq.empty()) {
      // one group = the oldest request's k, in arrival order, up to kCoalesceMaxBatch queries
      group.clear();
      const uint32_t gk = s->bq.front()->k;
      size_t total = 0;
      for (auto it = s->bq.begin(); it != s->bq.end();) {
        if ((*it)->k == gk && total + (*it)->nq <= kCoalesceMaxBatch) {
          total += (*it)->nq;
          group.push_back(*it);
          it = s->bq.erase(it);
        } else {
          ++it;
        }
      }
      lk.unlock();
      int rc;
      if (group.size() == 1) {
        ehx_space::KnnReq* r = group[0];
        rc = knn_host_direct(s, r->nq, r->q, gk, r->ids, r->dist, r->cnt);
      } else {
        q.resize(total * s->dims);
        ids.resize(total * gk);
        dist.resize(total * gk);
        cnt.resize(total);
        size_t off = 0;
        for (auto* r : group) {
          memcpy(q.data() + off * s->dims, r->q, r->nq * s->dims * sizeof(float));
          off += r->nq;
        }
        rc = knn_host_direct(s, total, q.data(), gk, ids.data(), dist.data(), cnt.data());
        off = 0;
        for (auto* r : group) {
          if (rc == EHX_OK) {
            memcpy(r->ids, ids.data() + off * gk, r->nq * gk * sizeof(uint64_t));
            memcpy(r->dist, dist.data() + off * gk, r->nq * gk * sizeof(float));
            memcpy(r->cnt, cnt.data() + off, r->nq * sizeof(uint32_t));
          }
          off += r->nq;
        }
        s->n_coalesced_batches += 1;
        s->n_coalesced_queries += total;
      }
      lk.lock();
      for (auto* r : group) {
        r->rc = rc;
        if (rc) snprintf(r->err, sizeof(r->err), "%s", g_err);
        r->done = true;
      }
      s->bq_cv.notify_all();
    }
    s->bq_leader = false;
    s->bq_cv.notify_all();  // a waiter whose request is still queued takes over
  }
  lk.unlock();
  if (me.rc) snprintf(g_err, sizeof(g_err), "%s", me.err);
  return me.rc;
}

int ehx_knn_keys(ehx_space* s, size_t n_queries, const float* queries, uint32_t k, uint64_t* out_ids,
                 float* out_dist, uint32_t* out_count, char* key_arena, size_t arena_cap, uint64_t* key_off) {
  if (!key_off || (!key_arena && arena_cap)) return fail(EHX_EINVAL, "NULL argument");
  int rc = ehx_knn(s, n_queries, queries, k, out_ids, out_dist, out_count);
  if (rc) return rc;
  std::shared_lock<std::shared_mutex> rl(s->mu);
  if (s->dropped) return fail(EHX_ENOTFOUND, "Not found");
  uint64_t off = 0;
  std::string key;
  for (size_t i = 0; i < n_queries; ++i) {
    for (uint32_t j = 0; j < k; ++j) {
      key_off[i * k + j] = off;
      if (j < out_count[i] && key_for_id(s, out_ids[i * k + j], &key) == EHX_OK) {
        if (off + key.size() > arena_cap) return fail(EHX_ERANGE, "key arena too small");
        memcpy(key_arena + off, key.data(), key.size());
        off += key.size();
      }
    }
  }
  key_off[n_queries * k] = off;
  return EHX_OK;
}

int ehx_knn_by_key(ehx_space* s, const char* key, size_t klen, uint32_t k, uint64_t* out_ids, float* out_dist,
                   uint32_t* out_count) {
  if (!valid_space(s) || !key || !out_count) return fail(EHX_EINVAL, "NULL argument");
  if (k > EHX_MAX_K_PAGED) return fail(EHX_EUNSUPPORTED, "k=%u exceeds %u", k, EHX_MAX_K_PAGED);  // (before k + 1, before any allocation)
  uint64_t id;
  std::vector<float> v(s->dims);
  {
    std::shared_lock<std::shared_mutex> rl(s->mu);
    if (s->dropped || lookup_key(s, key, klen, &id)) return fail(EHX_ENOTFOUND, "Not found");
  }
  int rc = ehx_get_by_id(s, id, v.data());  // Version::get(key), server.cc:195
  if (rc) return rc;
  const uint32_t kk = k + 1;               // server.cc:198
  std::vector<uint64_t> ids(kk);
  std::vector<float> dist(kk);
  uint32_t cnt = 0;
  if ((rc = ehx_knn(s, 1, v.data(), kk, ids.data(), dist.data(), &cnt))) return rc;
  // server.cc:205-207: erase own key if present, else drop the last
  uint32_t o = 0;
  bool removed = false;
  for (uint32_t j = 0; j < cnt; ++j) {
    if (!removed && ids[j] == id) {
      removed = true;
      continue;
    }
    if (o < k) {
      if (out_ids) out_ids[o] = ids[j];
      if (out_dist) out_dist[o] = dist[j];
      ++o;
    }
  }
  if (!removed && cnt == kk && o == k) { /* last one already dropped by the o<k bound */ }
  *out_count = o;
  return EHX_OK;
}

int ehx_knn_by_key_keys(ehx_space* s, const char* key, size_t klen, uint32_t k, uint64_t* out_ids, float* out_dist,
                        uint32_t* out_count, char* key_arena, size_t arena_cap, uint64_t* key_off) {
  if (!key_off || !out_count || (!key_arena && arena_cap)) return fail(EHX_EINVAL, "NULL argument");
  if (k > EHX_MAX_K_PAGED) return fail(EHX_EUNSUPPORTED, "k=%u exceeds %u", k, EHX_MAX_K_PAGED);
  std::vector<uint64_t> ids(k ? k : 1);
  int rc = ehx_knn_by_key(s, key, klen, k, ids.data(), out_dist, out_count);
  if (rc) return rc;
  std::shared_lock<std::shared_mutex> rl(s->mu);
  if (s->dropped) return fail(EHX_ENOTFOUND, "Not found");
  uint64_t off = 0;
  std::string nk;
  for (uint32_t j = 0; j < k; ++j) {
    key_off[j] = off;
    if (j < *out_count) {
      // (a returned row without a key cannot be answered as "": the RPC would hand back an empty neighbour key)
      if (key_for_id(s, ids[j], &nk) != EHX_OK)
        return fail(EHX_EINTERNAL, "row %llu was returned by the search but has no key", (unsigned long long)ids[j]);
      if (off + nk.size() > arena_cap) return fail(EHX_ERANGE, "key arena too small");
      memcpy(key_arena + off, nk.data(), nk.size());
      off += nk.size();
    }
    if (out_ids) out_ids[j] = j < *out_count ? ids[j] : ~0ull;
  }
  key_off[k] = off;
  return EHX_OK;
}

int ehx_merge_topk_strided_device(void* stream, size_t n_queries, uint32_t k, uint32_t n_lists, const uint64_t* d_ids,
                                  size_t ids_stride, const float* d_dist, size_t dist_stride, const uint32_t* d_count,
                                  size_t count_stride, uint64_t* d_out_ids, float* d_out_dist,
                                  uint32_t* d_out_count) {
  if (n_queries == 0 || k == 0) return EHX_OK;
  if (k > 64 && n_lists > 64) return fail(EHX_EUNSUPPORTED, "merging k > 64 takes at most 64 lists (%u)", n_lists);
  if (!d_ids || !d_dist || !d_out_ids || !d_out_dist) return fail(EHX_EINVAL, "NULL device pointer");
  if (ids_stride % 8 || dist_stride % 4 || count_stride % 4) return fail(EHX_EINVAL, "misaligned list stride");
  int rc = ehx_init(nullptr, 0);
  if (rc) return rc;
  HIP_TRY(launch_merge_lists(d_ids, d_dist, d_count, (uint32_t)n_queries, k, n_lists, d_out_ids, d_out_dist,
                             d_out_count, (hipStream_t)stream, ids_stride, dist_stride, count_stride));
  return EHX_OK;
}

int ehx_merge_topk_device(void* stream, size_t n_queries, uint32_t k, uint32_t n_lists, const uint64_t* d_ids,
                          const float* d_dist, const uint32_t* d_count, uint64_t* d_out_ids, float* d_out_dist,
                          uint32_t* d_out_count) {
  return ehx_merge_topk_strided_device(stream, n_queries, k, n_lists, d_ids, n_queries * k * sizeof(uint64_t), d_dist,
                                       n_queries * k * sizeof(float), d_count, n_queries * sizeof(uint32_t),
                                       d_out_ids, d_out_dist, d_out_count);
}

}  // extern "C"
