// Graph mode (hnswlib addPoint / updatePoint / searchKnn re-laid-out for the GPU): insertion rounds, update-in-place
// repair, and the search pipeline that chains the k_graph.hip / k_insert.hip kernels.
#include "ehx_internal.h"

namespace ehx_impl {

// ---- graph mode: GPU-side insertion of rows [id0, id0+count) (already in HBM, stats computed) ----
// hnswlib addPoint semantics (index.cc:36).  batch == 1: strictly sequential (the reference's
// mutex-serialised order); batch > 1: rounds of concurrent inserts against the graph as it was before
// the round (the analogue of hnswlib's multi-threaded add_items).
int graph_ensure_arrays(ehx_space* s) {
  const uint32_t M0 = 2 * s->params.M;
  if (s->g_cap_rows >= s->cap && s->dAdj0) return EHX_OK;
  HIP_TRY(hipDeviceSynchronize());
  uint32_t* na = nullptr;
  uint32_t* nu = nullptr;
  HIP_TRY(hipMalloc((void**)&na, s->cap * M0 * sizeof(uint32_t)));
  HIP_TRY(hipMalloc((void**)&nu, s->cap * sizeof(uint32_t)));
  HIP_TRY(hipMemset(na, 0xFF, s->cap * M0 * sizeof(uint32_t)));
  HIP_TRY(hipMemset(nu, 0xFF, s->cap * sizeof(uint32_t)));
  if (s->dAdj0 && s->g_n) {
    HIP_TRY(hipMemcpy(na, s->dAdj0, s->g_n * M0 * sizeof(uint32_t), hipMemcpyDeviceToDevice));
    HIP_TRY(hipMemcpy(nu, s->dUpStart, s->g_n * sizeof(uint32_t), hipMemcpyDeviceToDevice));
  }
  HIP_TRY(hipStreamSynchronize(nullptr));  // (fills and copies above ran on the NULL stream; ours are non-blocking)
  if (s->dAdj0) (void)hipFree(s->dAdj0);
  if (s->dUpStart) (void)hipFree(s->dUpStart);
  s->dAdj0 = na;
  s->dUpStart = nu;
  s->g_cap_rows = s->cap;
  return EHX_OK;
}

int graph_ensure_lists(ehx_space* s, uint64_t lists) {
  if (lists <= s->g_lists_cap && s->dUpLists) return EHX_OK;
  uint64_t want = s->g_lists_cap ? s->g_lists_cap : 1024;
  while (want < lists) want *= 2;
  HIP_TRY(hipDeviceSynchronize());
  uint32_t* nl = nullptr;
  HIP_TRY(hipMalloc((void**)&nl, want * s->params.M * sizeof(uint32_t)));
  HIP_TRY(hipMemset(nl, 0xFF, want * s->params.M * sizeof(uint32_t)));
  if (s->dUpLists && s->g_lists_used)
    HIP_TRY(hipMemcpy(nl, s->dUpLists, s->g_lists_used * s->params.M * sizeof(uint32_t), hipMemcpyDeviceToDevice));
  HIP_TRY(hipStreamSynchronize(nullptr));
  if (s->dUpLists) (void)hipFree(s->dUpLists);
  s->dUpLists = nl;
  s->g_lists_cap = want;
  return EHX_OK;
}

// hnswlib addPoint for rows [id0, id0 + count), already in HBM.  Rounds of P rows (P = 1: hnswlib's sequential
// insertion, the oracle's graph; P > 1: the analogue of its multi-threaded add_items) — and NO host work between a
// round's kernels: the levels of all rows are drawn up front (the generator's sequence does not depend on the graph),
// so the entry point and top level of every round are known to the host in advance; the search kernel writes the new
// nodes' own lists and registers the reverse links per adjacency list on the device, the link kernel applies them.
// The whole build is enqueued on the space's stream and waited for once.  (Round 2 paid three stream synchronisations,
// a std::map regrouping on the host, five small uploads and a 5-GB bitmap memset per round.)
int graph_insert(ehx_space* s, uint64_t id0, uint64_t count, uint32_t batch) {
  if (count == 0) return EHX_OK;
  if (id0 != s->g_n)
    return fail(EHX_EUNSUPPORTED, "graph mode: rows must be inserted in id order (graph covers %llu, next row %llu)",
                (unsigned long long)s->g_n, (unsigned long long)id0);
  const uint32_t M = s->params.M, M0 = 2 * M;
  if (M0 > 64 || M < 2) return fail(EHX_EUNSUPPORTED, "M=%u not supported by the insertion kernels (2M <= 64)", M);
  uint32_t efc = s->params.ef_construction > M ? s->params.ef_construction : M;  // max(efC, M)
  if (efc > 2048) return fail(EHX_EUNSUPPORTED, "ef_construction=%u exceeds 2048", efc);
  int rc;
  if ((rc = graph_ensure_arrays(s))) return rc;
  if (!s->level_rng_seeded) {
    s->level_rng.seed((unsigned)s->params.seed);
    s->level_rng_seeded = true;
  }
  const double mult = 1.0 / log(1.0 * M);
  hipStream_t st = s->stream;
  const uint64_t end = id0 + count;
  // ---- levels of every new row (getRandomLevel: -log(U(0,1)) * mult, a fresh distribution object per draw) ----
  // Nothing of this call is committed (level generator, h_levels, g_lists_used) before every allocation it needs has
  // succeeded: a call that fails for memory leaves the space exactly as it found it, and a retry draws the same levels.
  const std::default_random_engine rng_before = s->level_rng;
  auto undo = [&](int code) {
    s->level_rng = rng_before;
    return code;
  };
  std::vector<int32_t> h_lv(count);
  std::vector<uint32_t> h_upstart(count);
  uint64_t new_lists = 0;
  int top = s->g_n ? s->g_maxlevel : 0;
  for (uint64_t i = 0; i < count; ++i) {
    std::uniform_real_distribution<double> distribution(0.0, 1.0);
    const int level = (int)(-log(distribution(s->level_rng)) * mult);
    h_lv[i] = level;
    h_upstart[i] = level > 0 ? (uint32_t)(s->g_lists_used + new_lists) : 0xFFFFFFFFu;
    new_lists += (uint64_t)level;
    if (level > top) top = level;
  }
  if ((rc = graph_ensure_lists(s, s->g_lists_used + new_lists))) return undo(rc);
  if ((rc = s->dInsLevels.ensure(count))) return undo(rc);
  // ---- round schedule ----
  const uint64_t round_cap = batch > 1 ? batch : 4096;
  // Rows of one round do not see each other, so a round never exceeds a small share of the graph it joins: 1/128, at
  // most `round_cap` rows — and 1/256 when the graph stays small (below 128 Ki nodes after this call: there every node is
  // an early node, and hnswlib-python's add_items with 64 threads is blind to only 64 / n of the graph).  Measured
  // against the oracle's sequentially built graphs at equal ef (tests/test_graph_scale.py, recall@10 over 4096 queries,
  // worst ef; profiles/r03_*_graph_scale_report*.jsonl): share 1/16 — 20 k x 768 Gaussian rows -0.009, 200 k x 768
  // -0.0015; 1/64 — -0.005 and -0.001, but 200 k x 768 STRUCTURED rows (bench.py's manifold data, where recall is
  // 0.97 and neighbours are real) -0.0052; 1/128 — structured -0.0017; 1/256 — -0.0013.  A round costs ~2 ms however
  // few rows it holds (one wave's ef_construction search is a millisecond of dependent steps), so the small shares are
  // paid once, while the graph is small: 2 M x 768 takes 18.4 s with 1/16 and 19.2 s with 1/64.
  // EHX_BUILD_DIV overrides the share (A/B runs).
  const uint64_t div_env = env().build_div;
  const uint64_t div = div_env >= 2 ? div_env : (end < (128u << 10) ? 256 : 128);
  auto round_size = [&](uint64_t g_n, uint64_t left) {
    uint64_t P = 1;
    if (batch != 1 && g_n >= 64) {
      P = g_n / div;
      if (P > round_cap) P = round_cap;
      if (P < 1) P = 1;
    }
    return P > left ? left : P;
  };
  uint64_t n_rounds = 0, max_P = 1;
  for (uint64_t g = s->g_n, pos = id0; pos < end; ++n_rounds) {
    const uint64_t P = g ? round_size(g, end - pos) : 1;
    if (P > max_P) max_P = P;
    g += P;
    pos += P;
  }
  const uint32_t vis_words = (uint32_t)((s->cap + 31) / 32);
  const uint32_t vislog_cap = 32768;
  const uint64_t max_pairs = max_P * (uint64_t)(top + 1) * M;
  if (max_pairs >= 0xFFFFFFFFull) return undo(fail(EHX_EUNSUPPORTED, "graph build: round too large"));
  // (the bitmaps are zero when allocated and every search clears the bits it set: no per-round memset)
  if ((rc = s->dVisited.ensure(max_P * vis_words, true))) return undo(rc);
  if ((rc = s->dInsVislog.ensure(max_P * (uint64_t)vislog_cap))) return undo(rc);
  if ((rc = s->dLinkHead.ensure(s->cap + s->g_lists_cap, true))) return undo(rc);  // all zero between rounds
  if ((rc = s->dLinkNext.ensure(max_pairs))) return undo(rc);
  if ((rc = s->dLinkTouched.ensure(max_pairs))) return undo(rc);
  if ((rc = s->dLinkCount.ensure(n_rounds))) return undo(rc);
  // ---- commit: from here on the rows are on their way into the graph ----
  s->g_lists_used += new_lists;
  s->h_levels.insert(s->h_levels.end(), h_lv.begin(), h_lv.end());
  // the new nodes' up_start entries and levels (their adjacency rows are still all-0xFF; nothing reaches a node
  // before the round that links it)
  HIP_TRY(hipMemcpyAsync(s->dUpStart + id0, h_upstart.data(), count * sizeof(uint32_t), hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(s->dInsLevels.p, h_lv.data(), count * sizeof(int32_t), hipMemcpyHostToDevice, st));
  if (s->vis_dirty) {  // a search that clears its bitmaps before its kernel left them marked
    HIP_TRY(hipMemsetAsync(s->dVisited.p, 0, s->dVisited.n * sizeof(uint32_t), st));
    s->vis_dirty = false;
  }
  HIP_TRY(hipMemsetAsync(s->dLinkCount.p, 0, n_rounds * sizeof(uint32_t), st));
  InsertArgs a{};  // (zeroed: a null link_head / sel switches those outputs off in the kernels)
  a.X = (s->x_half || s->x_perm) ? nullptr : s->xf32();  // (graph kernels read the search copy; X: fp32 ablation builds only)
  a.Xs = s->dXs;
  a.inv_norm = s->dInv;
  a.xscale = (s->x_perm && s->metric == EHX_METRIC_COSINE) ? s->dInv : nullptr;
  a.adj0 = s->dAdj0;
  a.up_start = s->dUpStart;
  a.up_lists = s->dUpLists;
  a.visited = s->dVisited.p;
  a.vislog = s->dInsVislog.p;
  a.new_ids = nullptr;
  a.sel = nullptr;
  a.ef = efc;
  a.dims = s->dims;
  a.ld = s->ld;
  a.M = M;
  a.M0 = M0;
  a.vis_words = vis_words;
  a.vislog_cap = vislog_cap;
  a.metric = s->metric;
  a.exclude_self = 0;
  a.head_rows = (uint32_t)s->cap;
  a.link_head = s->dLinkHead.p;
  a.link_next = s->dLinkNext.p;
  a.link_touched = (uint2*)s->dLinkTouched.p;
  // EHX_BUILD_TRACE=1: progress to stderr (costs a stream synchronisation every 128 rounds)
  const bool trace = env().build_trace;
  const auto t_build0 = std::chrono::steady_clock::now();
  uint64_t pos = id0, round = 0;
  while (pos < end) {
    if (s->g_n == 0) {  // very first node: becomes the entry point, nothing to link
      s->g_entry = (uint32_t)pos;
      s->g_maxlevel = h_lv[0];
      s->g_n = 1;
      pos += 1;
      round += 1;
      continue;
    }
    const uint64_t P = round_size(s->g_n, end - pos);
    a.id0 = (uint32_t)pos;
    a.new_levels = s->dInsLevels.p + (pos - id0);
    a.max_sel_levels = (uint32_t)s->g_maxlevel + 1;
    a.entry_point = s->g_entry;
    a.max_level = s->g_maxlevel;
    a.link_count = s->dLinkCount.p + round;
    HIP_TRY(launch_insert_search(a, (uint32_t)P, st));
    const uint64_t pairs = P * a.max_sel_levels * M;
    HIP_TRY(launch_insert_link_dev(a, (uint32_t)std::min<uint64_t>(pairs, 32768), st));
    // entry point / top level (hnswlib: a node with a higher level becomes the entry point)
    for (uint64_t i = 0; i < P; ++i) {
      const int lv = h_lv[pos - id0 + i];
      if (lv > s->g_maxlevel) {
        s->g_entry = (uint32_t)(pos + i);
        s->g_maxlevel = lv;
      }
    }
    s->g_n += P;
    pos += P;
    round += 1;
    if (trace && ((round & 127) == 0 || pos >= end)) {
      HIP_TRY(hipStreamSynchronize(st));
      fprintf(stderr, "[ehx build] round %llu of %llu, rows %llu, %.1f s\n", (unsigned long long)round,
              (unsigned long long)n_rounds, (unsigned long long)s->g_n,
              std::chrono::duration<double>(std::chrono::steady_clock::now() - t_build0).count());
    }
  }
  HIP_TRY(hipStreamSynchronize(st));
  // A bulk build gives its scratch back: one visited bitmap per insertion in flight is cap / 8 bytes each — 5.1 GB for
  // rounds of 4096 rows on a 10 M-row index, four times what a 1024-query search batch needs (it re-allocates its own,
  // zeroed, at its first call: ~1 ms).  Streamed Sets (small calls) keep theirs.
  // (EHX_BUILD_SCRATCH_KEEP=<bytes>: what a build may keep, whatever its size — tests release at small sizes with 0)
  const long long keep_env = env().build_scratch_keep;
  const bool give_back = keep_env >= 0 ? s->dVisited.n * sizeof(uint32_t) > (unsigned long long)keep_env
                                       : (end - id0 >= 65536 && s->dVisited.n * sizeof(uint32_t) > (1ull << 30));
  if (give_back) {
    s->dVisited.release();
    s->dInsVislog.release();
    s->dLinkNext.release();
    s->dLinkTouched.release();
    s->vis_dirty = false;
  }
  return EHX_OK;
}

// ---- graph mode: hnswlib updatePoint(data, id, 1.0) for a row overwritten in place (index.cc:21-36:
// an existing key keeps its label and addPoint takes its update branch) ----
int graph_update(ehx_space* s, uint32_t id) {
  if (id >= s->g_n) return EHX_OK;
  if (s->g_entry == id && s->g_n == 1) return EHX_OK;
  const uint32_t M = s->params.M, M0 = 2 * M;
  const uint32_t efc = s->params.ef_construction > M ? s->params.ef_construction : M;
  hipStream_t st = s->stream;
  const int level = s->h_levels[id];
  int rc;
  InsertArgs a{};  // (zeroed: a null link_head / sel switches those outputs off in the kernels)
  a.X = (s->x_half || s->x_perm) ? nullptr : s->xf32();  // (graph kernels read the search copy; X: fp32 ablation builds only)
  a.Xs = s->dXs;
  a.inv_norm = s->dInv;
  a.xscale = (s->x_perm && s->metric == EHX_METRIC_COSINE) ? s->dInv : nullptr;
  a.adj0 = s->dAdj0;
  a.up_start = s->dUpStart;
  a.up_lists = s->dUpLists;
  a.ef = efc;
  a.dims = s->dims;
  a.ld = s->ld;
  a.M = M;
  a.M0 = M0;
  a.metric = s->metric;
  a.entry_point = s->g_entry;
  a.max_level = s->g_maxlevel;
  a.exclude_self = 1;
  auto read_list = [&](uint32_t node, int layer, std::vector<uint32_t>* out) -> int {
    const uint32_t width = layer == 0 ? M0 : M;
    uint32_t buf[64];
    const uint32_t* src;
    if (layer == 0) {
      src = s->dAdj0 + (size_t)node * M0;
    } else {
      uint32_t us = 0;
      HIP_TRY(hipMemcpy(&us, s->dUpStart + node, 4, hipMemcpyDeviceToHost));
      src = s->dUpLists + ((size_t)us + (uint32_t)(layer - 1)) * M;
    }
    HIP_TRY(hipMemcpy(buf, src, width * 4, hipMemcpyDeviceToHost));
    out->clear();
    for (uint32_t j = 0; j < width && buf[j] != 0xFFFFFFFFu; ++j) out->push_back(buf[j]);
    return EHX_OK;
  };
  // part 1: the one-hop neighbours re-select their links among {id} u one-hop u two-hop
  std::vector<uint32_t> one, two, h_neigh, h_off, h_cand;
  for (int layer = 0; layer <= level; ++layer) {
    if ((rc = read_list(id, layer, &one))) return rc;
    if (one.empty()) continue;
    std::set<uint32_t> sCand;
    sCand.insert(id);
    for (uint32_t o : one) {
      sCand.insert(o);
      if ((rc = read_list(o, layer, &two))) return rc;
      for (uint32_t t : two) sCand.insert(t);
    }
    h_neigh.assign(one.begin(), one.end());
    std::sort(h_neigh.begin(), h_neigh.end());
    h_neigh.erase(std::unique(h_neigh.begin(), h_neigh.end()), h_neigh.end());
    h_off.clear();
    h_cand.clear();
    for (uint32_t ng : h_neigh) {
      h_off.push_back((uint32_t)h_cand.size());
      for (uint32_t c : sCand)
        if (c != ng) h_cand.push_back(c);
    }
    h_off.push_back((uint32_t)h_cand.size());
    const uint32_t n_items = (uint32_t)h_neigh.size();
    if ((rc = s->dItemTgt.ensure(n_items))) return rc;
    if ((rc = s->dItemOff.ensure(n_items + 1))) return rc;
    if ((rc = s->dItemIds.ensure(h_cand.size() ? h_cand.size() : 1))) return rc;
    HIP_TRY(hipMemcpyAsync(s->dItemTgt.p, h_neigh.data(), n_items * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(s->dItemOff.p, h_off.data(), (n_items + 1) * 4, hipMemcpyHostToDevice, st));
    if (!h_cand.empty())
      HIP_TRY(hipMemcpyAsync(s->dItemIds.p, h_cand.data(), h_cand.size() * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(launch_update_neigh(a, n_items, s->dItemTgt.p, layer, s->dItemOff.p, s->dItemIds.p, st));
    HIP_TRY(hipStreamSynchronize(st));
  }
  // part 2: repairConnectionsForUpdate = search from the entry point with the new vector, drop the
  // node itself from the results, reconnect with isUpdate semantics
  const uint32_t vis_words = (uint32_t)((s->cap + 31) / 32);
  const uint32_t vislog_cap = 32768;
  const uint32_t max_sel_levels = (uint32_t)s->g_maxlevel + 1;
  if ((rc = s->dInsIds.ensure(1))) return rc;
  if ((rc = s->dInsLevels.ensure(1))) return rc;
  if ((rc = s->dInsSel.ensure((size_t)max_sel_levels * (1 + M)))) return rc;
  if ((rc = s->dVisited.ensure(vis_words, true))) return rc;
  if ((rc = s->dInsVislog.ensure(vislog_cap))) return rc;
  const int32_t lv32 = level;
  HIP_TRY(hipMemcpyAsync(s->dInsIds.p, &id, 4, hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(s->dInsLevels.p, &lv32, 4, hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemsetAsync(s->dVisited.p, 0, vis_words * sizeof(uint32_t), st));
  a.visited = s->dVisited.p;
  a.vislog = s->dInsVislog.p;
  a.new_ids = s->dInsIds.p;
  a.new_levels = s->dInsLevels.p;
  a.sel = s->dInsSel.p;
  a.vis_words = vis_words;
  a.vislog_cap = vislog_cap;
  a.max_sel_levels = max_sel_levels;
  HIP_TRY(launch_insert_search(a, 1, st));
  std::vector<uint32_t> h_sel((size_t)max_sel_levels * (1 + M));
  HIP_TRY(hipMemcpyAsync(h_sel.data(), s->dInsSel.p, h_sel.size() * 4, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  std::vector<uint32_t> h_tgt, h_kind, h_ioff, h_inc;
  std::vector<int32_t> h_tlevel;
  for (int l = 0; l <= level; ++l) {
    const uint32_t* o = &h_sel[(size_t)l * (1 + M)];
    const uint32_t c = o[0];
    if (c == 0) continue;  // level skipped by hnswlib: lists untouched
    h_tgt.push_back(id);
    h_tlevel.push_back(l);
    h_kind.push_back(1u);
    h_ioff.push_back((uint32_t)h_inc.size());
    for (uint32_t j = 0; j < c; ++j) h_inc.push_back(o[1 + j]);
    for (uint32_t j = 0; j < c; ++j) {  // reverse links, in hnswlib's order (selectedNeighbors order)
      h_tgt.push_back(o[1 + j]);
      h_tlevel.push_back(l);
      h_kind.push_back(2u);
      h_ioff.push_back((uint32_t)h_inc.size());
      h_inc.push_back(id);
    }
  }
  h_ioff.push_back((uint32_t)h_inc.size());
  const uint32_t n_items = (uint32_t)h_tgt.size();
  if (n_items) {
    if ((rc = s->dItemTgt.ensure(n_items))) return rc;
    if ((rc = s->dItemLevel.ensure(n_items))) return rc;
    if ((rc = s->dItemKind.ensure(n_items))) return rc;
    if ((rc = s->dItemOff.ensure(n_items + 1))) return rc;
    if ((rc = s->dItemIds.ensure(h_inc.size()))) return rc;
    HIP_TRY(hipMemcpyAsync(s->dItemTgt.p, h_tgt.data(), n_items * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(s->dItemLevel.p, h_tlevel.data(), n_items * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(s->dItemKind.p, h_kind.data(), n_items * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(s->dItemOff.p, h_ioff.data(), (n_items + 1) * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(s->dItemIds.p, h_inc.data(), h_inc.size() * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(launch_insert_link(a, n_items, s->dItemTgt.p, s->dItemLevel.p, s->dItemKind.p, s->dItemOff.p,
                               s->dItemIds.p, st));
    HIP_TRY(hipStreamSynchronize(st));
  }
  return EHX_OK;
}

// graph pipeline: prepared queries -> zero visited bitmaps -> one-wave-per-query search
// `one` (optional): ONE query per call in ONE launch — the raw query sits in host-visible memory (one->q_host), the
// kernel prepares it itself, writes ids / distances / count into host-visible memory (d_ids / d_dist / d_count then point
// there) and publishes one->seq in one->done_flag; no prepare launch, no timing events (nothing between the host's
// launch and the wave's first instruction but the runtime).
int knn_graph_locked(ehx_space* s, hipStream_t st, size_t nq, const float* d_queries, uint32_t k, uint64_t* d_ids,
                     float* d_dist, uint32_t* d_count, const GraphOneLaunch* one) {
  if (s->poisoned.load())
    return fail(EHX_EINTERNAL, "graph space: an in-place overwrite failed half way (rows left in raw order); drop and rebuild it");
  if (s->g_n != s->n)
    return fail(EHX_EUNSUPPORTED,
                "graph mode: the graph covers %llu of %llu rows (rows were written while graph building was "
                "switched off, build_batch = 0xFFFFFFFF: import the graph with ehx_graph_import)",
                (unsigned long long)s->g_n, (unsigned long long)s->n);
  uint32_t ef = s->params.ef > k ? s->params.ef : k;  // searchKnn: max(ef_, k)
  if (ef > 4096) return fail(EHX_EUNSUPPORTED, "ef=%u exceeds 4096", ef);
  const uint32_t q_rows = (uint32_t)nq;
  int rc;
  if ((rc = s->dQ.ensure((size_t)q_rows * s->ld))) return rc;
  const uint32_t vis_words = (uint32_t)((s->n + 31) / 32);
  // the bitmaps are all-zero between kernels (every kernel that marks rows clears them again): zeroed once, on
  // allocation
  if ((rc = s->dVisited.ensure((size_t)nq * vis_words, true))) return rc;
  const bool use_vislog = env().graph_vislog;  // (EHX_GRAPH_VISLOG=0: per-batch memset of the bitmaps instead, A/B runs)
  // Measured (r02, batch 1024, memset inside the timed region; gpurun_out of scripts/gpu_session_n.sh): the memset
  // costs n/8 bytes per query, streamed; the log costs one store per visited row plus one RANDOM 4-byte store per row
  // when the query clears its words — ~27 ef of them.  6.25 M x 128: ef 50 log 0.41 / memset 0.47 ms, ef 200 1.11 /
  // 1.11, ef 800 4.27 / 3.89; 2 M x 768: ef 100 2.06 / 2.05, ef 400 6.99 / 6.81; small bitmaps (1 M x 128): the
  // memset is nearly free.  Hence: the log when the bitmaps are large AND the index has more than 32 000 rows per ef.
  // (one launch: always the log — the wave clears the handful of words it marked; a memset would be a second command)
  const bool log_now = one ? true
                           : use_vislog && (size_t)nq * vis_words * sizeof(uint32_t) >= (192u << 20) &&
                                 s->n >= (uint64_t)32000 * ef;
  const uint32_t vislog_cap = log_now ? 48u * ef + 256u : 0u;
  if ((rc = s->dInsVislog.ensure((size_t)nq * (vislog_cap ? vislog_cap : 1u)))) return rc;
  if (!s->dGraphCounters) {
    HIP_TRY(hipMalloc((void**)&s->dGraphCounters, kGraphCounters * sizeof(unsigned long long)));
    HIP_TRY(hipMemset(s->dGraphCounters, 0, kGraphCounters * sizeof(unsigned long long)));
  }
  {
    int rcw = wait_searches_in_flight(s, st);
    if (rcw) return rcw;
  }
  // timing events: batch 0 after a reset (outside the ring's mean) and every EHX_STATS_EVERY-th batch (ehx_internal.h)
  const uint64_t batch_no = one ? 0 : s->g_batches++;
  const bool in_ring = !one && (batch_no % env().stats_every) == env().stats_every - 1u;
  const bool timed = !one && (in_ring || batch_no == 0);
  if (!one) {
    if (timed) HIP_TRY(hipEventRecord(s->ev[0], st));
    HIP_TRY(launch_prep_queries(d_queries, (uint32_t)nq, s->dims, s->ld, q_rows, s->metric, s->dQ.p, st));
  }
  GraphArgs a;
  a.Q = s->dQ.p;
  a.X = (s->x_half || s->x_perm) ? nullptr : s->xf32();  // (graph kernels read the search copy; X: fp32 ablation builds only)
  a.Xs = s->dXs;
  a.inv_norm = s->dInv;
  a.xscale = (s->x_perm && s->metric == EHX_METRIC_COSINE) ? s->dInv : nullptr;
  a.adj0 = s->dAdj0;
  a.up_start = s->dUpStart;
  a.up_lists = s->dUpLists;
  a.visited = s->dVisited.p;
  a.vislog = s->dInsVislog.p;
  a.vislog_cap = vislog_cap;
  a.out_ids = d_ids;
  a.out_dist = d_dist;
  a.out_count = d_count;
  a.counters = s->dGraphCounters;
  a.nq = (uint32_t)nq;
  a.k = k;
  a.ef = ef;
  a.ef_cap = ef;
  a.n = (uint32_t)s->n;
  a.dims = s->dims;
  a.ld = s->ld;
  a.M = s->params.M;
  a.M0 = 2 * s->params.M;
  a.vis_words = vis_words;
  a.entry_point = s->g_entry;
  a.max_level = s->g_maxlevel;
  a.metric = s->metric;
  // The wide walk expands `width` entries of a list of ef at once: with a short list that is a large share of it, and the
  // walk fetches rows the strict order would have pruned (200 k x 768 structured rows, ef 10 / 20 / 40, rows fetched against
  // the strict walk: width 2 +9 / +4 / +2 %, width 4 +30 / +16 / +7 %; profiles/r06_b_wide_gate.jsonl; 6.25 M x 128, ef 50:
  // width 4 fetches 9 % more rows than width 2 for the same kernel time, profiles/r06_d_graph_6250k128.jsonl).  Hence
  // two expansions per step from ef = 16 on, four from ef = 64 on; ef < 16 — the reference's default ef = 10 among them —
  // is always walked strictly.
  a.width = env().graph_width ? env().graph_width : (s->params.search_width > 1 ? s->params.search_width : 1);
  if (a.width >= 4 && ef < 64) a.width = 2;
  if (a.width >= 2 && ef < 16) a.width = 1;
  if (one) {
    a.q_raw = one->q_host;
    a.done_flag = one->done_flag;
    a.seq = one->seq;
    if (s->vis_dirty) HIP_TRY(hipMemsetAsync(s->dVisited.p, 0, s->dVisited.n * sizeof(uint32_t), st));
    s->vis_dirty = false;
    HIP_TRY(launch_graph_search(a, st));
    s->n_queries += nq;
    return EHX_OK;
  }
  hipEvent_t* pr = in_ring ? s->ring[s->ring_count % ehx_space::kRing] : &s->ev[1];   // (batch 0: ev[1] / ev[2])
  if (timed) {
    s->g_timed_valid = false;
    HIP_TRY(hipEventRecord(pr[0], st));
  }
  // (inside the timed kernel region: clearing the bitmaps is part of what a batch costs, log or memset)
  if (log_now && s->vis_dirty)  // (the whole buffer: an earlier, larger batch may have marked words beyond this one's)
    HIP_TRY(hipMemsetAsync(s->dVisited.p, 0, s->dVisited.n * sizeof(uint32_t), st));
  else if (!log_now)
    HIP_TRY(hipMemsetAsync(s->dVisited.p, 0, (size_t)nq * vis_words * sizeof(uint32_t), st));
  s->vis_dirty = !log_now;
  HIP_TRY(launch_graph_search(a, st));
  if (timed) {
    HIP_TRY(hipEventRecord(pr[1], st));
    s->scan_ev[0] = pr[0];
    s->scan_ev[1] = pr[1];
    if (in_ring) s->ring_count++;
  }
  HIP_TRY(hipEventRecord(s->ev[3], st));
  s->ev3_stream = st;
  s->end_sampled = true;
  if (timed) {
    HIP_TRY(hipEventRecord(s->ev_end, st));
    s->g_timed_valid = true;
  }
  s->ev_valid = true;
  s->ev_seq = ++s->ev_counter;
  s->n_queries += nq;
  return EHX_OK;
}

}  // namespace ehx_impl
