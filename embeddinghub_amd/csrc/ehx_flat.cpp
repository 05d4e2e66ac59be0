// The exact flat chain: scan plans, the fp32 / fp16 / int8 matrix-core pipelines with their certified re-rank, the
// exhaustive canonical pass, and the engine fall-through of one device batch (knn_device_locked).
#include "ehx_internal.h"

namespace ehx_impl {

struct ScanPlan {
  uint32_t q_tiles, q_rows, n_tiles, n_chunks, tiles_per_chunk, kprime, xcd_map, grid;
};

// plan one scan pass over `n_tiles` row tiles
ScanPlan plan_scan(uint32_t nq, uint32_t n_tiles, uint32_t k, int n_cus) {
  ScanPlan p;
  p.q_tiles = (nq + kTileQ - 1) / kTileQ;
  p.q_rows = p.q_tiles * kTileQ;
  p.n_tiles = n_tiles;
  p.kprime = k + 8;  // EHX_MAX_K + 8 = 56 < kCandSlots: a compacted candidate list always has free slots
  // one persistent workgroup per CU: grid ~= n_cus, split as q_tiles x n_chunks
  uint32_t chunks = (uint32_t)n_cus / p.q_tiles;
  if (chunks < 1) chunks = 1;
  if (chunks >= 8) chunks &= ~7u;
  if (chunks > p.n_tiles) chunks = p.n_tiles ? p.n_tiles : 1;
  p.n_chunks = chunks;
  p.tiles_per_chunk = p.n_tiles ? (p.n_tiles + chunks - 1) / chunks : 0;
  p.grid = p.q_tiles * p.n_chunks;
  p.xcd_map = (p.n_chunks % 8 == 0) ? 1u : 0u;
  return p;
}

// one flat pipeline: prepared queries -> scan -> merge -> canonical re-rank.
//   f16 = false: the fp32 MFMA scan (k_flat8.hip), exact on its own.
//   f16 = true : the fp16 MFMA filter scan (k_flat16.hip); per-query certification flags land in
//                s->dUflags and the caller re-runs the unflagged remainder through the fp32 scan.
int flat_pass(ehx_space* s, hipStream_t st, size_t nq, const float* d_queries, uint32_t k, uint64_t* d_ids,
              float* d_dist, uint32_t* d_count, bool f16, bool count_stats) {
  Engine& E = engine();
  // A cascade of scan passes over growing row ranges (one tile per workgroup, then x8 per pass): after
  // every pass the per-workgroup candidate lists are merged into the query's running best-64 and its
  // k'-th best key — an upper bound of the final k'-th best — becomes the threshold the next pass starts
  // from.  A pass over 8x the rows seen so far appends only ~7 k' candidates per query, so nearly every
  // tile epilogue stays on its branch-free fast path; a single pass would have every workgroup warm its
  // thresholds up from +inf (~k' ln(rows/k') appends per list).
  const uint32_t tile_rows = f16 ? kTileRows16 : kTileRows;
  const uint32_t n_tiles = (uint32_t)((s->n + tile_rows - 1) / tile_rows);
  const uint32_t lpc = f16 ? 2u : scan_lists_per_chunk();
  struct Pass {
    uint32_t tile0;
    ScanPlan plan;
  };
  // Filter scan: a SAMPLE pass first — the first 8 tiles (2048 rows) are scanned in dump mode (all scores to
  // HBM, no candidate lists) and sample_select turns them into the k'-th best score per query, so not even
  // the first real pass has to warm its lists up from +inf (which costs ~3 list compactions per list).
  constexpr uint32_t kSampleTiles = 8;
  const bool sample = f16 && n_tiles >= 256;
  std::vector<Pass> passes;
  {
    const ScanPlan whole = plan_scan((uint32_t)nq, n_tiles, k, E.n_cus);
    uint32_t done = 0;
    if (sample) {
      // the filter scan gets its first thresholds from the sample pass below, so its cascade can start
      // wide (128 tiles) and grow x16: three scan launches at 10 M rows
      uint32_t cum = 128;
      while (cum * 2 < n_tiles) {
        passes.push_back({done, plan_scan((uint32_t)nq, cum - done, k, E.n_cus)});
        done = cum;
        cum *= 16;
      }
    } else if (lpc == 2 && n_tiles >= 16 * whole.n_chunks) {
      uint32_t cum = whole.n_chunks;  // pass 0: one tile per workgroup
      while (cum * 2 < n_tiles) {
        passes.push_back({done, plan_scan((uint32_t)nq, cum - done, k, E.n_cus)});
        done = cum;
        cum *= 8;
      }
    }
    passes.push_back({done, plan_scan((uint32_t)nq, n_tiles - done, k, E.n_cus)});
  }
  ScanPlan p = passes.back().plan;  // (q_tiles, q_rows, kprime are the same for every pass)
  if (f16) {
    // the filter keeps k' = k + 22 candidates (<= 56): the certification needs the k'-th lower bound to
    // clear the k-th exact distance by the fp16 error bound, so it wants more slack than the fp32 scan
    const uint32_t kp = k + 22 > 56 ? (k + 8 > 56 ? k + 8 : 56) : k + 22;
    for (auto& ps : passes) ps.plan.kprime = kp;
    p.kprime = kp;
  } else if (s->dims > 1024) {
    // fp32 scan at large d: the certification margin grows like d * 2^-24 (cert_margin) while the gap between the
    // k-th and the k'-th best of isotropic data shrinks like ln(k'/k) / sqrt(d): widen k' (up to the 56 a
    // 64-slot list allows) so that typical data still certifies instead of falling to the exhaustive pass
    const double grow = std::exp(std::min(4.0, 5.3e-7 * std::pow((double)s->dims, 1.5)));
    uint32_t kp = (uint32_t)std::ceil((double)k * grow);
    kp = std::min<uint32_t>(56, std::max<uint32_t>(k + 8, kp));
    for (auto& ps : passes) ps.plan.kprime = kp;
    p.kprime = kp;
  }
  uint32_t lists_total = 0, grid_max = 0;  // every pass reuses the same list slots
  for (auto& ps : passes) {
    lists_total = std::max(lists_total, ps.plan.n_chunks * lpc);
    grid_max = std::max(grid_max, ps.plan.grid);
  }
  int rc;
  if ((rc = s->dQ.ensure((size_t)p.q_rows * s->ld))) return rc;
  if ((rc = s->dCand.ensure((size_t)grid_max * 512 * kCandSlots))) return rc;
  if ((rc = s->dPart.ensure((size_t)p.q_rows * lists_total * p.kprime))) return rc;
  if ((rc = s->dMerged.ensure((size_t)p.q_rows * 64))) return rc;
  if ((rc = s->dGthr.ensure((size_t)p.q_rows + 8))) return rc;  // +8: instrumentation slots of profiling builds
  if (!s->dUncert) {
    HIP_TRY(hipMalloc((void**)&s->dUncert, 2 * sizeof(unsigned long long)));  // [0] uncertified, [1] scan error
    HIP_TRY(hipMemset(s->dUncert, 0, 2 * sizeof(unsigned long long)));
  }
  if ((rc = s->dUflags.ensure(p.q_rows))) return rc;
  if (!s->dUncert16) {
    HIP_TRY(hipMalloc((void**)&s->dUncert16, sizeof(unsigned long long)));
    HIP_TRY(hipMemset(s->dUncert16, 0, sizeof(unsigned long long)));
  }
  if (f16) {
    if ((rc = s->dQ16.ensure(scanq16_halves(p.q_rows, s->ld16)))) return rc;
    if ((rc = s->dQgamma.ensure(p.q_rows))) return rc;
    if ((rc = s->dQuv.ensure(p.q_rows))) return rc;
    if (sample && (rc = s->dSample.ensure((size_t)kSampleTiles * kTileRows16 * p.q_rows))) return rc;
  }
  // scratch buffers are shared by all callers: order this pipeline after the previous one even
  // when it was enqueued on a different stream
  {
    int rcw = wait_searches_in_flight(s, st);
    if (rcw) return rcw;
  }
  HIP_TRY(hipEventRecord(s->ev[0], st));
  HIP_TRY(launch_prep_queries(d_queries, (uint32_t)nq, s->dims, s->ld, p.q_rows, s->metric, s->dQ.p, st));
  if (f16)
    HIP_TRY(launch_prep_queries16(d_queries, (uint32_t)nq, s->dims, s->ld16, p.q_rows, s->metric, s->dQ16.p,
                                  s->dQgamma.p, s->dQuv.p, st));
  if (s->n == 0) {
    // empty space: every query returns count 0
    HIP_TRY(hipMemsetAsync(s->dMerged.p, 0xFF, (size_t)p.q_rows * 64 * sizeof(uint64_t), st));
    HIP_TRY(hipEventRecord(s->ev[1], st));
    HIP_TRY(hipEventRecord(s->ev[2], st));
  } else {
    ScanArgs a;
    a.Q = s->dQ.p;
    a.X = s->dX;
    a.x_half = (uint32_t)s->x_half;
    a.rowp = s->dRowp;
    a.cand = s->dCand.p;
    a.part = s->dPart.p;
    a.n = (uint32_t)s->n;
    a.ld = s->ld;
    a.q_tiles = p.q_tiles;
    a.kprime = p.kprime;
    a.lists_total = lists_total;
    a.err = (uint32_t*)(s->dUncert + 1);
    a.gthr = (unsigned long long*)s->dGthr.p;
    ScanArgs16 h;
    h.Q = s->dQ16.p;
    h.X = s->dX16;
    h.rowp = s->dRowp16;
    h.qgamma = s->dQgamma.p;
    h.eps = scan16_eps(s->dims);
    h.cos = s->metric == EHX_METRIC_COSINE;
    h.cand = a.cand;
    h.part = a.part;
    h.n = a.n;
    h.ld = s->ld16;
    h.q_tiles = a.q_tiles;
    h.kprime = a.kprime;
    h.lists_total = lists_total;
    h.err = a.err;
    h.gthr = a.gthr;
    auto scan = [&](const ScanPlan& pl, uint32_t tile0, uint32_t list0) -> hipError_t {
      if (f16) {
        h.tile0 = tile0;
        h.n_tiles = pl.n_tiles;
        h.n_chunks = pl.n_chunks;
        h.tiles_per_chunk = pl.tiles_per_chunk;
        h.xcd_map = pl.xcd_map;
        h.list0 = list0;
        return launch_flat_scan16(h, st);
      }
      a.tile0 = tile0;
      a.n_tiles = pl.n_tiles;
      a.n_chunks = pl.n_chunks;
      a.tiles_per_chunk = pl.tiles_per_chunk;
      a.xcd_map = pl.xcd_map;
      a.list0 = list0;
      return launch_flat_scan(a, st);
    };
    HIP_TRY(hipMemsetAsync(s->dGthr.p, 0xFF, (size_t)p.q_rows * sizeof(uint64_t), st));
    hipEvent_t* pr = s->ring[s->ring_count % ehx_space::kRing];
    HIP_TRY(hipEventRecord(s->ev[1], st));
    HIP_TRY(hipEventRecord(pr[0], st));
    if (sample) {
      ScanPlan sp = plan_scan((uint32_t)nq, kSampleTiles, k, E.n_cus);
      sp.kprime = p.kprime;
      h.dump = s->dSample.p;
      HIP_TRY(scan(sp, 0, 0));
      h.dump = nullptr;
      HIP_TRY(launch_sample_select(s->dSample.p, kSampleTiles * kTileRows16, p.q_rows, (uint32_t)nq, p.kprime,
                                   (unsigned long long*)s->dGthr.p, st));
    }
    for (size_t i = 0; i < passes.size(); ++i) {
      const bool last = i + 1 == passes.size();
      HIP_TRY(scan(passes[i].plan, passes[i].tile0, 0));
      if (last) {  // (the final merge is outside the timed scan phase)
        HIP_TRY(hipEventRecord(pr[1], st));
        HIP_TRY(hipEventRecord(s->ev[2], st));
        s->ring_count++;
      }
      HIP_TRY(launch_flat_merge(s->dPart.p, (uint32_t)nq, passes[i].plan.n_chunks * lpc, p.kprime, s->dMerged.p, st,
                                lists_total, i > 0, last ? nullptr : (unsigned long long*)s->dGthr.p));
    }
  }
  RerankArgs r;
  r.Q = s->dQ.p;
  r.X = s->dX;
  r.x_half = (uint32_t)s->x_half;
  r.inv_norm = s->dInv;
  r.merged = s->dMerged.p;
  r.out_ids = d_ids;
  r.out_dist = d_dist;
  r.out_count = d_count;
  r.n_uncertified = s->dUncert16;  // verdict counter of this pass (the caller reads and clears it)
  r.nq = (uint32_t)nq;
  r.k = k;
  r.kprime = p.kprime;
  r.n = (uint32_t)s->n;
  r.dims = s->dims;
  r.ld = s->ld;
  r.metric = s->metric;
  if (f16) r.quv = s->dQuv.p;
  r.max_sumsq = s->dMaxSumsq;
  r.uncert_flags = s->dUflags.p;
  HIP_TRY(launch_rerank(r, st));
  HIP_TRY(hipEventRecord(s->ev[3], st));
  s->ev3_stream = st;
  s->end_sampled = false;
  s->ev_valid = true;
  s->ev_seq = ++s->ev_counter;
  if (count_stats) {
    s->n_queries += nq;
    s->n_dist += (uint64_t)nq * s->n;
    // SURVEY §8d brute force bytes per batch: N*d*s + B*d*4 + B*k*12 (s = bytes per element the scan reads)
    s->bytes_algo += s->n * s->dims * (uint64_t)(f16 ? 2 : s->esz) + (uint64_t)nq * s->dims * 4ull +
                     (uint64_t)nq * k * 12ull;
  }
  s->n_rerank += (uint64_t)nq * p.kprime;
  return EHX_OK;
}

// which scan engine answers first on this space right now: EHX_ENGINE_* (include/ehx.h)
int resolve_engine(const ehx_space* s) {
  if (s->params.mode != EHX_MODE_FLAT || s->scan_sel == EHX_SCAN_F32 || s->n == 0) return EHX_ENGINE_F32;
  if (s->scan_sel == EHX_SCAN_AUTO && s->has8 && s->h_unsafe8 == 0 && s->n >= s->i8_min_rows) return EHX_ENGINE_I8;
  if (s->has16 && s->h_unsafe == 0) return EHX_ENGINE_F16;
  return EHX_ENGINE_F32;
}

// The int8 filter pipeline (k_flati8.hip, k_select.hip): prepared queries -> sample pass (first thresholds) ->
// cascade of collect passes, x4 in rows, each followed by select256 (running best 256 + the next threshold) ->
// rerank256 (canonical distances of the k' = 128 best lower bounds, top-k, certificate).  Per-query verdicts land in
// s->dUflags / s->dUncert16 like those of flat_pass.
// `set`: which of the space's two scratch sets (ehx_space::I8Set) this batch runs in; the caller holds that set's mutex.
int flat_pass8(ehx_space* s, int set, hipStream_t st, size_t nq, const float* d_queries, uint32_t k, uint64_t* d_ids,
               float* d_dist, uint32_t* d_count, bool count_stats, uint32_t* kprime_used) {
  Engine& E = engine();
  ehx_space::I8Set& sc = s->i8set[set];
  const uint32_t n_tiles = (uint32_t)((s->n + kTileRows16 - 1) / kTileRows16);
  // The cascade's shape.  Shards of at least 2048 tiles (524 288 rows): a first pass of 128 tiles that aims for 64 keys
  // per query, then x8 in rows per pass; smaller ones: 512 tiles, min(512, 2 k') keys, x4 (one or two passes).  Round 5,
  // same box, 60 batches each (profiles/r05_i_ab_cascade.jsonl): 1.25 M x 768 1.190 -> 1.146 ms per batch, 10 M x 768
  // 6.64 -> 6.58, 6.25 M x 128 1.154 -> 1.139 (256 / x8 / 64: 12.5 M x 1536 fp16 18.7 -> 18.4); at 200 000 x 768 the
  // second pass the small first pass brings costs more than its tighter threshold saves (scan phase 0.248 -> 0.274).
  // A loose first pass is the expensive one: its threshold comes from a 2048-row sample and every key it lets through
  // is a trip through the epilogue's slow path.  EHX_I8_FIRST_TILES / EHX_I8_GROWTH / EHX_I8_FIRST_KEYS override.
  const bool big = n_tiles >= 2048;
  const uint32_t growth = env().i8_growth ? env().i8_growth : (big ? 8u : 4u);
  const double safety = [] {
    // rank of the next pass's threshold = 256 x (share of the rows seen) x safety.  2: the last pass of a 10 M-row batch
    // runs under the 138th best of the first 27 % (about 510 rows below it overall: the list still fills to 256, the
    // certificate's floor is unchanged) instead of the 256th (950 rows): fewer alarms, fewer keys — 7.64 -> 7.46 ms per
    // batch, 0 fallbacks over 25 batches; 1.5: 7.44; 1: a query falls to the next engine (profiles/r03_e_i8_safety_sweep.jsonl)
    return env().i8_safety;  // (EHX_I8_SAFETY; 1e9: always the 256th best)
  }();
  // EHX_I8_SYNC = N > 0: lock-step by tile of the query-tile workgroups that stream one row chunk, tolerance N tiles
  // (k_flati8.hip); 0 / unset: off.
  const int sync_mode = env().i8_sync;
  const bool use_sync = sync_mode > 0;
  constexpr uint32_t kSampleTiles = 8;
  // The int8 bound leaves ~60-75 rows per query ON AVERAGE that it cannot exclude from the top-10 at 10 M rows
  // (scripts/studies/int8_filter_bound.py), with a heavy tail — a query whose 10th neighbour is unusually far has
  // several times as many — and a query whose list is too short costs a whole scan by the next engine.  So the
  // threshold rank is the full width of the running list; the re-rank reads only as many candidates as it needs.
  // The list is 256 keys wide to begin with.  Longer rows leave more survivors (the bound is ~0.02 in dot units whatever
  // the dimension, while the spread of the dot products shrinks like 1/sqrt(d)): at 12.5 M x 1536 a fifth of the
  // queries needed more than 256 candidates and went to the next engine, which doubled the batch time.  A space whose
  // batches keep losing queries that way doubles its list (knn_device_locked), up to kMerged8Max.
  const uint32_t width = s->i8_width.load(std::memory_order_relaxed);  // (read once: another batch may widen it meanwhile)
  const long kprime_env = env().i8_kprime;
  // How many candidates a query keeps is what the scan's epilogue pays for (every key collected is a trip through its
  // slow path, and the waves of a workgroup wait for each other at every stage: 1.25 M x 768 collected 470 keys per
  // query, 70 % of the epilogue's tests alarmed).  The rows a query cannot exclude grow with the index (60-75 on average
  // at 10 M x 768, fewer on a shard of 1 M), so the list's LOGICAL length k' follows the row count — 128 below 4 M rows,
  // else the full width (64 was tried: no query lost at 1 - 1.25 M x 768 in 50 batches, 3 of 256 at 100 k x 768); rows
  // of 1024 dims and more always get the full width (the bound is ~1.3e-2 in dot units whatever d while the scores'
  // spread shrinks like 1/sqrt(d): 18 000 x 2048 needs its 256) — and, like the width, doubles when queries lose their
  // certificate because the list was too short (knn_device_locked).
  // Short rows need fewer still: the bound is a smaller share of the scores' spread (0.15 sigma at d = 128 against 0.36
  // at d = 768), and on short rows the hit path is what a tile's time is made of (two stages of matrix work per tile at
  // d = 128).  Measured, 46 000 queries each, 0 fallbacks (profiles/r04_p_kprime_short_rows.jsonl): 6.25 M x 128 k' 256 /
  // 128 / 64 = 1.380 / 1.277 / 1.238 ms per batch, 1 M x 128 0.447 (128) / 0.414 (64); 4 M x 384 and 10 M x 256 are fine
  // with 128 (1.762 against 1.825, 2.620 against 2.722 ms) and lose queries by the hundred with 64.
  uint32_t kp_auto = s->n >= 4000000 ? 256u : 128u;
  if (s->dims <= 128) kp_auto = 64u;
  else if (s->dims < 512) kp_auto = 128u;
  if (s->dims >= 1024) kp_auto = width;
  const uint32_t kp_want = std::max(kp_auto, s->i8_kprime_min.load(std::memory_order_relaxed));
  const uint32_t kprime = kprime_env >= 64 ? (uint32_t)std::min<long>(kprime_env, (long)width) : std::min(kp_want, width);
  s->i8_kprime_last.store(kprime, std::memory_order_relaxed);
  if (kprime_used) *kprime_used = kprime;
  struct Pass {
    uint32_t tile0;
    ScanPlan plan;
  };
  // First pass: up to 512 tiles (131 072 rows) under a threshold taken from the sample at a LOW rank, chosen so that
  // the pass collects ~1000 keys per query (any threshold is sound, see sample_select256_kernel); then x4 in rows per
  // pass under the 256th best so far.
  const uint32_t kFirstTiles = env().i8_first_tiles ? env().i8_first_tiles : (big ? 128u : 512u);
  std::vector<Pass> passes;
  {
    uint32_t done = 0, cum = kFirstTiles;
    while ((uint64_t)cum * 2 < n_tiles) {
      passes.push_back({done, plan_scan((uint32_t)nq, cum - done, k, E.n_cus)});
      done = cum;
      cum *= growth;
    }
    passes.push_back({done, plan_scan((uint32_t)nq, n_tiles - done, k, E.n_cus)});
  }
  // rank of the threshold the select after pass i publishes for pass i + 1: the kprime-th best is always valid; while
  // only a share f of the rows has been seen, the final kprime-th best is expected near rank kprime * f of the prefix, so
  // rank kprime * f * safety (>= 16) is a much tighter threshold that is still above it — fewer keys collected, fewer
  // epilogue alarms in the middle passes.  Sound whatever happens (the certificate uses the smallest threshold ever
  // applied, qparams.w).
  auto rank_after = [&](size_t i) -> uint32_t {
    if (i + 1 >= passes.size()) return kprime;
    const double f = (double)((uint64_t)(passes[i + 1].tile0) * kTileRows16) / (double)s->n;
    return (uint32_t)std::min<double>(kprime, std::max<double>(16.0, std::ceil(kprime * f * safety)));
  };
  // The first pass runs under a threshold taken from the sample at a LOW rank, chosen for the number of keys the pass
  // should collect: a single pass has to fill the list with room to spare (2 x its logical length — every key collected
  // is a trip through the epilogue's slow path, and on a 20 000-row space 4 x meant a hit in 2.6 % of all pairs); with more
  // passes to come it only has to deliver the next threshold's rank (twice over, at least min(512, 2 k') keys: round 2
  // collected 1024 and spent more than half of the first pass in the epilogue's slow path).
  const uint64_t first_rows = (uint64_t)passes.front().plan.n_tiles * kTileRows16;
  const uint64_t first_keys_env = env().i8_first_keys;  // (EHX_I8_FIRST_KEYS: keys per query the first pass aims for)
  const uint64_t first_floor = first_keys_env ? first_keys_env : (big ? 64ull : std::min<uint64_t>(512, 2ull * kprime));
  const uint64_t first_keys = std::min<uint64_t>(
      2048, passes.size() == 1 ? 2ull * kprime : std::max<uint64_t>(first_floor, 2ull * rank_after(0)));
  const uint32_t sample_rank =
      (uint32_t)std::min<uint64_t>(64, std::max<uint64_t>(8, first_keys * kSampleTiles * kTileRows16 / first_rows));
  const ScanPlan p = passes.back().plan;  // (q_tiles, q_rows are the same for every pass)
  uint32_t grid_max = 0, chunks_max = 0;
  for (auto& ps : passes) {
    grid_max = std::max(grid_max, ps.plan.grid);
    chunks_max = std::max(chunks_max, ps.plan.n_chunks);
  }
  if (chunks_max > 256) return fail(EHX_EINTERNAL, "scan plan with %u chunks", chunks_max);
  int rc;
  if ((rc = sc.dQ.ensure((size_t)p.q_rows * s->ld))) return rc;
  if ((rc = sc.dQ8.ensure(scanq8_bytes(p.q_rows, s->ld8)))) return rc;
  if ((rc = sc.dQp8.ensure(p.q_rows))) return rc;
  if ((rc = sc.dQuv.ensure(p.q_rows))) return rc;
  if ((rc = sc.dThr8.ensure(p.q_rows))) return rc;
  if ((rc = sc.dSample8.ensure((size_t)kSampleTiles * kTileRows16 * p.q_rows))) return rc;
  if ((rc = sc.dCnt.ensure(8, true))) return rc;   // (the set's own: this function runs outside the pipeline lock too)
  if ((rc = sc.dPool.ensure((size_t)p.q_rows * kPoolCap))) return rc;
  if ((rc = sc.dMerged8.ensure((size_t)p.q_rows * width))) return rc;
  if ((rc = sc.dI8Ctl.ensure((size_t)p.q_rows * 2 + kSyncWordsI8))) return rc;
  if ((rc = sc.dUflags.ensure(p.q_rows))) return rc;
  if (!sc.dUncert) {
    HIP_TRY(hipMalloc((void**)&sc.dUncert, sizeof(unsigned long long)));
    HIP_TRY(hipMemset(sc.dUncert, 0, sizeof(unsigned long long)));
    HIP_TRY(hipHostMalloc((void**)&sc.hUncertPin, sizeof(unsigned long long), hipHostMallocDefault));
  }
  uint32_t* pool_cnt = sc.dI8Ctl.p;
  uint32_t* ovf = sc.dI8Ctl.p + p.q_rows;
  uint32_t* sync = sc.dI8Ctl.p + 2 * (size_t)p.q_rows;
  if (!sc.ev[0]) {
    for (auto& e : sc.ev) HIP_TRY(hipEventCreate(&e));
    for (auto& pr2 : sc.ring)
      for (auto& e : pr2) HIP_TRY(hipEventCreate(&e));
    for (auto& e : sc.first_pair) HIP_TRY(hipEventCreate(&e));
    HIP_TRY(hipEventCreateWithFlags(&sc.verdict, hipEventBlockingSync | hipEventDisableTiming));
  }
  // (a caller's stream other than the space's own: searches already in flight there and here finish first)
  if ((rc = wait_searches_in_flight(s, st))) return rc;
  const uint64_t batch_no = sc.batches++;
  const bool in_ring = (batch_no % env().stats_every) == env().stats_every - 1u;
  const bool timed = in_ring || batch_no == 0;   // (the first batch after a reset is a timed one, outside the ring's mean)
  if (timed) HIP_TRY(hipEventRecord(sc.ev[0], st));
  HIP_TRY(launch_prep_queries_i8(d_queries, (uint32_t)nq, s->dims, s->ld, s->ld8, p.q_rows, s->metric, sc.dQ.p,
                                 sc.dQ8.p, sc.dQp8.p, sc.dQuv.p, sc.dThr8.p, sc.dI8Ctl.p, st));
  ScanArgsI8 a;
  a.Q = sc.dQ8.p;
  a.X = s->dX8;
  a.rowp = s->dRowp8;
  a.tilep = s->dTilep8;
  a.tileg = s->dTileg8;
  a.perm = s->dPerm8;
  a.qparams = sc.dQp8.p;
  a.thr = sc.dThr8.p;
  a.cand = sc.dCnt.p;
  a.pool = sc.dPool.p;
  a.pool_cnt = pool_cnt;
  a.ovf = ovf;
  a.pool_cap = kPoolCap;
  a.n = (uint32_t)s->n;
  a.ld = s->ld8;
  a.q_tiles = p.q_tiles;
  a.skew = env().i8_skew;
  // (cosine / inner product: B_r is one constant, every margin 0; L2^2 on normalised rows: no tile has a margin worth the
  // epilogue's extra permute and multiply-add per query block — 6.25 M x 128: 1.02 -> 1.07 ms per batch with them)
  a.group_b = s->metric == EHX_METRIC_L2SQ && s->h_margin8 > 0 && env().i8_groupb ? 1u : 0u;
  auto scan = [&](const ScanPlan& pl, uint32_t tile0) -> hipError_t {
    a.tile0 = tile0;
    a.n_tiles = pl.n_tiles;
    a.n_chunks = pl.n_chunks;
    a.tiles_per_chunk = pl.tiles_per_chunk;
    a.xcd_map = pl.xcd_map;
    return launch_flat_scan_i8(a, st);
  };
  hipEvent_t* pr = in_ring ? sc.ring[sc.ring_count % 64] : sc.first_pair;
  // (one record per mark: sc.ev[1] / ev[2] — "scan start / end of the LAST batch" for ehx_stats — are the ring's own events
  // of this batch; a second marker packet at the same place cost ~5 us of queue time each, twice per batch)
  if (timed) {
    HIP_TRY(hipEventRecord(pr[0], st));
    sc.last_scan[0] = sc.last_scan[1] = nullptr;   // (a pass that fails half way leaves no half pair for ehx_stats)
    sc.timed_valid = false;
  }
  {  // sample pass: lower bounds of the first 2048 rows -> thr[q] = the k'-th best of them
    ScanPlan sp = plan_scan((uint32_t)nq, kSampleTiles, k, E.n_cus);
    a.dump = sc.dSample8.p;
    a.sync = nullptr;
    HIP_TRY(scan(sp, 0));
    a.dump = nullptr;
    HIP_TRY(launch_sample_select256(sc.dSample8.p, kSampleTiles * kTileRows16, p.q_rows, (uint32_t)nq, sample_rank,
                                    sc.dThr8.p, st));
  }
  for (size_t i = 0; i < passes.size(); ++i) {
    const bool last = i + 1 == passes.size();
    a.sync = nullptr;
    if (use_sync && passes[i].plan.xcd_map && p.q_tiles > 1 && passes[i].plan.tiles_per_chunk >= 4 &&
        passes[i].plan.n_chunks * 4u <= kSyncWordsI8) {   // (four progress words per chunk: ADVICE r05)
      a.sync = sync;
      a.sync_tol = (uint32_t)sync_mode;
      if (i > 0) HIP_TRY(hipMemsetAsync(sync, 0, kSyncWordsI8 * sizeof(uint32_t), st));
    }
    HIP_TRY(scan(passes[i].plan, passes[i].tile0));
    if (last && timed) {  // (the last select and the re-rank are outside the timed scan phase, like flat_pass's final merge)
      HIP_TRY(hipEventRecord(pr[1], st));
      sc.last_scan[0] = pr[0];   // both marks of THIS batch, set together
      sc.last_scan[1] = pr[1];
      if (in_ring) sc.ring_count++;
    }
    HIP_TRY(launch_select256(sc.dPool.p, pool_cnt, kPoolCap, (uint32_t)nq, rank_after(i), sc.dMerged8.p, width, i > 0,
                             sc.dThr8.p, sc.dQp8.p, st));
  }
  Rerank256Args r;
  r.Q = sc.dQ.p;
  r.X = s->dX;
  r.x_half = (uint32_t)s->x_half;
  r.inv_norm = s->dInv;
  r.merged = sc.dMerged8.p;
  r.width = width;
  r.ovf = ovf;
  r.quv = sc.dQuv.p;
  r.qparams = sc.dQp8.p;
  r.max_sumsq = s->dMaxSumsq;
  r.out_ids = d_ids;
  r.out_dist = d_dist;
  r.out_count = d_count;
  r.n_uncertified = sc.dUncert;
  r.uncert_flags = sc.dUflags.p;
  r.nq = (uint32_t)nq;
  r.k = k;
  r.kprime = kprime;
  r.n = (uint32_t)s->n;
  r.dims = s->dims;
  r.ld = s->ld;
  r.metric = s->metric;
  HIP_TRY(launch_rerank256(r, st));
  if (env().i8_count) {  // diagnosis builds (-DEHX_I8_COUNT=1): the scan's epilogue counters of this batch
    unsigned long long c[8] = {0};
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipMemcpy(c, sc.dCnt.p, sizeof(c), hipMemcpyDeviceToHost));
    fprintf(stderr, "[i8 count] tests %llu alarms %llu row-block alarms %llu trips %llu (cumulative)\n", c[0], c[1], c[2], c[3]);
  }
  if (env().i8_debug) {  // diagnosis only: what the uncertified queries of this batch look like
    HIP_TRY(hipStreamSynchronize(st));
    std::vector<uint32_t> fl(nq), ov(nq);
    std::vector<float4> qp(nq);
    std::vector<float2> uv(nq);
    std::vector<uint64_t> mg(nq * width);
    std::vector<float> od(nq * k);
    HIP_TRY(hipMemcpy(fl.data(), sc.dUflags.p, nq * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(ov.data(), ovf, nq * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(qp.data(), sc.dQp8.p, nq * sizeof(float4), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(uv.data(), sc.dQuv.p, nq * sizeof(float2), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(mg.data(), sc.dMerged8.p, nq * width * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(od.data(), d_dist, nq * k * 4, hipMemcpyDeviceToHost));
    int shown = 0;
    for (size_t q = 0; q < nq && shown < 6; ++q) {
      if (!fl[q]) continue;
      ++shown;
      auto S = [&](int i) { return mg[q * width + i] == ~0ull ? INFINITY : ordered_to_f32((uint32_t)(mg[q * width + i] >> 32)); };
      fprintf(stderr, "[i8 debug] q=%zu ovf=%u tmin=%g S[0]=%g S[63]=%g S[127]=%g S[255]=%g kth_dist=%g u=%g v=%g\n", q, ov[q],
              qp[q].w, S(0), S(63), S(127), S(255), od[q * k + k - 1], uv[q].x, uv[q].y);
    }
  }
  HIP_TRY(hipEventRecord(sc.ev[3], st));
  sc.ev3_stream = st;
  sc.ev_valid = true;
  if (timed) {
    HIP_TRY(hipEventRecord(sc.ev[2], st));
    sc.timed_valid = true;
    sc.ev_seq = ++s->ev_counter;
  }
  if (count_stats) {
    s->n_queries += nq;
    s->n_dist += (uint64_t)nq * s->n;
    // SURVEY §8d brute force bytes per batch: N*d*s + B*d*4 + B*k*12 (s = 1: the int8 scan copy)
    s->bytes_algo += s->n * (uint64_t)s->dims + (uint64_t)nq * s->dims * 4ull + (uint64_t)nq * k * 12ull;
  }
  s->n_rerank += (uint64_t)nq * kprime;
  return EHX_OK;
}

// Exhaustive canonical pass: the canonical distance of every row for `nq` queries (k_flat.hip:
// exhaustive_kernel), merged and emitted through the re-rank with the certification switched off (the keys
// are exact).  Serves (a) queries no matrix-core scan can certify and (b) requests with k > EHX_MAX_K, which
// it answers in pages of 64 results (each page keeps the keys strictly above the previous page's last).
int exhaustive_pass(ehx_space* s, hipStream_t st, size_t nq, const float* d_queries, uint32_t k, uint64_t* d_ids,
                    float* d_dist, uint32_t* d_count) {
  // rows per workgroup: 8192 when there are queries enough to fill the chip; fewer queries get smaller blocks (down to
  // one 64-row step) so that about 2048 workgroups share the shard — the keys are exact whatever the partition
  const uint32_t kRowsPerBlock =
      (uint32_t)std::min<uint64_t>(8192, std::max<uint64_t>(64, ((uint64_t)s->n * nq / 2048 + 63) / 64 * 64));
  const uint32_t n_blocks = (uint32_t)((s->n + kRowsPerBlock - 1) / kRowsPerBlock);
  const uint32_t pages = (k + 63) / 64;
  int rc;
  if ((rc = s->dQ.ensure(nq * s->ld))) return rc;
  if ((rc = s->dPart.ensure(nq * n_blocks * 64))) return rc;
  if ((rc = s->dMerged.ensure(nq * 64))) return rc;
  if ((rc = s->dUflags.ensure(nq))) return rc;
  if (pages > 1 && (rc = s->dGthr.ensure(nq + 8))) return rc;
  if (!s->dUncert16) {
    HIP_TRY(hipMalloc((void**)&s->dUncert16, sizeof(unsigned long long)));
    HIP_TRY(hipMemset(s->dUncert16, 0, sizeof(unsigned long long)));
  }
  {
    int rcw = wait_searches_in_flight(s, st);
    if (rcw) return rcw;
  }
  HIP_TRY(hipEventRecord(s->ev[0], st));
  HIP_TRY(launch_prep_queries(d_queries, (uint32_t)nq, s->dims, s->ld, (uint32_t)nq, s->metric, s->dQ.p, st));
  HIP_TRY(hipEventRecord(s->ev[1], st));
  for (uint32_t pg = 0; pg < pages; ++pg) {
    const uint64_t* floor = pg ? s->dGthr.p : nullptr;
    HIP_TRY(launch_exhaustive(s->dQ.p, s->dX, s->x_half, s->dInv, (uint32_t)s->n, s->dims, s->ld, s->metric,
                              kRowsPerBlock, n_blocks, (uint32_t)nq, floor, s->dPart.p, st));
    HIP_TRY(launch_flat_merge(s->dPart.p, (uint32_t)nq, n_blocks, 64, s->dMerged.p, st, n_blocks));
    if (pg + 1 < pages) HIP_TRY(launch_set_floor(s->dMerged.p, (uint32_t)nq, s->dGthr.p, st));
    RerankArgs r;
    r.Q = s->dQ.p;
    r.X = s->dX;
    r.x_half = (uint32_t)s->x_half;
    r.inv_norm = s->dInv;
    r.merged = s->dMerged.p;
    r.out_ids = d_ids;
    r.out_dist = d_dist;
    r.out_count = d_count;
    r.n_uncertified = s->dUncert16;
    r.nq = (uint32_t)nq;
    r.k = std::min<uint32_t>(64, k - pg * 64);
    r.kprime = 64;
    r.n = (uint32_t)s->n;
    r.dims = s->dims;
    r.ld = s->ld;
    r.metric = s->metric;
    r.uncert_flags = s->dUflags.p;
    r.exact_keys = 1;
    r.out_stride = k;
    r.out_offset = pg * 64;
    HIP_TRY(launch_rerank(r, st));
    if (pg + 1 == pages) HIP_TRY(hipEventRecord(s->ev[2], st));
  }
  HIP_TRY(hipEventRecord(s->ev[3], st));
  s->ev3_stream = st;
  s->end_sampled = false;
  s->ev_valid = true;
  s->ev_seq = ++s->ev_counter;
  s->n_dist += (uint64_t)nq * s->n * pages;
  return EHX_OK;
}

// Adaptation of the int8 list after a batch of `nq` queries that ran with logical length `kprime`, lost `n_failed`
// queries to the next engine, `n_short` of them because their candidate LIST was too short.  Called for EVERY int8
// batch, clean ones included, from both paths (knn_device_locked; knn_host_direct's pipelined stage, which used to
// skip it for clean batches: its score never decayed, and two losing batches any distance apart widened the list).
// The list is too short for this data when batches keep losing queries to the next engine — which re-reads every
// row for them, nearly a batch's worth of time however few they are (12.5 M x 1536: 13 queries in 10 batches cost
// 45 % of the run).  Only queries whose LIST was the failing part count (the re-rank flags them 2: a pool overflow,
// exact ties at the threshold or lost candidates are not cured by width, and a width, once raised, stays).  A batch of
// at least 64 queries that loses more than 2 % of them that way widens the list at once; otherwise every losing
// batch adds 4 to a score that decays by 1 per clean batch, and 8 widens (two losing batches close together).
void i8_adapt(ehx_space* s, size_t nq, size_t n_failed, size_t n_short, uint32_t kprime) {
  std::lock_guard<std::mutex> l(s->i8_adapt_mu);
  if (n_short == 0) s->i8_fb_score = s->i8_fb_score ? s->i8_fb_score - 1 : 0;
  else s->i8_fb_score += 4;
  const uint32_t width = s->i8_width.load(std::memory_order_relaxed);
  if (!((nq >= 64 && n_short * 50 > nq) || s->i8_fb_score >= 8)) return;
  // (a batch that ran with an older, shorter list than the space has by now says nothing about the present one)
  if (kprime < std::min(width, std::max(s->i8_kprime_min.load(std::memory_order_relaxed), kprime))) {
    s->i8_fb_score = 0;
    return;
  }
  if (!(width < kMerged8Max || kprime < width)) return;
  if (kprime < width) s->i8_kprime_min.store(std::min(width, 2 * kprime), std::memory_order_relaxed);  // k' first
  else {
    s->i8_width.store(width * 2, std::memory_order_relaxed);
    s->i8_kprime_min.store(width * 2, std::memory_order_relaxed);
  }
  s->i8_fb_score = 0;
  if (env().i8_trace)
    fprintf(stderr, "[ehx i8] %zu of %zu queries uncertified (%zu by a short list): candidate list now %u of %u\n", n_failed,
            nq, n_short, std::max(s->i8_kprime_min.load(), kprime), s->i8_width.load());
}

// Device pipeline of a flat space: up to three stages, each run only for the queries the previous one
// could not certify, so the answer is always the exhaustive fp32 answer in the oracle's arithmetic:
//   0. int8 matrix-core filter scan + certified re-rank      (all queries; spaces with the int8 scan copy, >= i8_min_rows)
//   1. fp16 matrix-core filter scan + certified re-rank      (what stage 0 could not certify / spaces without it)
//   2. fp32 matrix-core scan + certified re-rank              (what stage 1 could not certify / fp32-only spaces)
//   3. canonical distance of every row                        (what stage 2 could not certify: near-ties finer
//                                                              than the certification margin; kMaxExhaustive
//                                                              queries per launch group, as many groups as needed)
// One host round trip (8 bytes) per stage to read its verdict.
// i8_failed (optional): the int8 stage of this very batch has already run — in one of the scratch sets, outside the
// pipeline lock (knn_host_direct) — and left the answers of every other query in the output arrays; these queries
// (i8_short of them because their candidate list was too short) continue with the next engine.
int knn_device_locked(ehx_space* s, hipStream_t st, size_t nq, const float* d_queries, uint32_t k,
                      uint64_t* d_ids, float* d_dist, uint32_t* d_count, const std::vector<uint32_t>* i8_failed,
                      size_t i8_short, uint32_t i8_kprime_in) {
  if (k == 0 || nq == 0) return EHX_OK;
  if (nq > (1u << 24)) return fail(EHX_EINVAL, "too many queries in one call: %zu", nq);
  if (s->params.mode == EHX_MODE_GRAPH) {
    // searchKnn(q, k) keeps max(ef, k) results and returns the k best (index.cc:41): any k the result list holds
    if (k > EHX_MAX_K_PAGED) return fail(EHX_EUNSUPPORTED, "graph mode: k=%u exceeds %u", k, EHX_MAX_K_PAGED);
    return knn_graph_locked(s, st, nq, d_queries, k, d_ids, d_dist, d_count);
  }
  if (k > EHX_MAX_K) {
    // beyond the candidate capacity of one scan pass: the exhaustive canonical pass, paged (exact, HBM-bound —
    // the whole shard is read once per page of 64 results and per query)
    if (k > EHX_MAX_K_PAGED) return fail(EHX_EUNSUPPORTED, "k=%u exceeds EHX_MAX_K_PAGED=%u", k, EHX_MAX_K_PAGED);
    if (s->n == 0) {
      HIP_TRY(hipMemsetAsync(d_count, 0, nq * sizeof(uint32_t), st));
      return EHX_OK;
    }
    int rc2 = exhaustive_pass(s, st, nq, d_queries, k, d_ids, d_dist, d_count);
    if (rc2) return rc2;
    s->n_queries += nq;
    s->n_exhaustive += nq;
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipMemsetAsync(s->dUncert16, 0, sizeof(unsigned long long), st));
    return EHX_OK;
  }
  // ONE query against a small shard — the reference's own usage: one NearestNeighbor RPC, one query (server.cc:172-210;
  // BASELINE configs[0]: 10 k x 128).  The matrix-core engines are built for batches: their dozen launches (sample pass,
  // cascade, selects, re-rank) take ~0.55 ms for a single query on 10 k rows, where the exhaustive canonical pass — the
  // oracle's arithmetic over every row, exact by construction, three launches — reads the rows once.  Concurrent single
  // queries never get here alone: ehx_knn coalesces them into device batches.  (EHX_SMALL_EXACT_BYTES=0 switches it off.)
  const uint64_t small_bytes = env().small_exact_bytes;
  if (nq == 1 && s->scan_sel == EHX_SCAN_AUTO && s->n > 0 && (uint64_t)s->n * s->ld * s->esz <= small_bytes) {
    int rc2 = exhaustive_pass(s, st, nq, d_queries, k, d_ids, d_dist, d_count);
    if (rc2) return rc2;
    s->n_queries += nq;
    s->n_exhaustive += nq;
    if (s->dUncert16) HIP_TRY(hipMemsetAsync(s->dUncert16, 0, sizeof(unsigned long long), st));
    return EHX_OK;   // (no wait here: the caller's copy-back or stream order is the wait)
  }
  constexpr size_t kMaxExhaustive = 32;
  enum { kI8, kFilter, kF32, kExhaustive };
  int rc;
  size_t n_short = 0;  // of the last stage's uncertified queries: those whose candidate LIST was too short (flag 2)
  uint32_t i8_kprime = i8_kprime_in;  // the k' this batch's int8 stage ran with
  // run one stage on `subset` (nullptr = every query); *unc = global indices it could not certify
  auto stage = [&](int kind, const std::vector<uint32_t>* subset, bool count_stats, std::vector<uint32_t>* unc) -> int {
    const size_t m = subset ? subset->size() : nq;
    const float* q = d_queries;
    uint64_t* oi = d_ids;
    float* od = d_dist;
    uint32_t* oc = d_count;
    if (subset) {
      if ((rc = s->dFbQ.ensure(m * s->dims))) return rc;
      if ((rc = s->dFbIds.ensure(m * k))) return rc;
      if ((rc = s->dFbDist.ensure(m * k))) return rc;
      if ((rc = s->dFbCnt.ensure(m))) return rc;
      if ((rc = s->dFbIdx.ensure(m))) return rc;
      // (the index list comes from pageable host memory: the runtime stages it before the call returns)
      HIP_TRY(hipMemcpyAsync(s->dFbIdx.p, subset->data(), m * sizeof(uint32_t), hipMemcpyHostToDevice, st));
      HIP_TRY(launch_gather_queries(d_queries, s->dFbIdx.p, (uint32_t)m, s->dims, s->dFbQ.p, st));
      q = s->dFbQ.p;
      oi = s->dFbIds.p;
      od = s->dFbDist.p;
      oc = s->dFbCnt.p;
    }
    // (the int8 stage runs in scratch set 0 here, held for the stage and its verdict: host batches may be using both sets
    // through knn_host_direct's pipelined path at the same time)
    std::unique_lock<std::mutex> set_lock(s->i8set[0].mu, std::defer_lock);
    if (kind == kI8) set_lock.lock();
    if (kind == kExhaustive) rc = exhaustive_pass(s, st, m, q, k, oi, od, oc);
    else if (kind == kI8) {   // (enqueued as one block, like a pipelined host batch's: per-batch scan windows stay clean)
      std::lock_guard<std::mutex> ql(s->i8_enqueue_mu);
      rc = flat_pass8(s, 0, st, m, q, k, oi, od, oc, count_stats, &i8_kprime);
    }
    else rc = flat_pass(s, st, m, q, k, oi, od, oc, kind == kFilter, count_stats);
    if (rc) return rc;
    unsigned long long* d_unc = kind == kI8 ? s->i8set[0].dUncert : s->dUncert16;
    const uint32_t* d_flags = kind == kI8 ? s->i8set[0].dUflags.p : s->dUflags.p;
    if (subset) {
      HIP_TRY(launch_scatter_results(oi, od, oc, s->dFbIdx.p, (uint32_t)m, k, d_ids, d_dist, d_count, st));
      HIP_TRY(hipEventRecord(s->ev[3], st));
      s->ev3_stream = st;
  s->end_sampled = false;
    }
    // verdict
    unc->clear();
    // (into PINNED host memory: a copy to pageable memory goes through a staging buffer and a copy kernel)
    if (!s->hUncertPin) HIP_TRY(hipHostMalloc((void**)&s->hUncertPin, sizeof(unsigned long long), hipHostMallocDefault));
    HIP_TRY(hipMemcpyAsync(s->hUncertPin, d_unc, sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    const unsigned long long n_unc = *s->hUncertPin;
#if defined(EHX_ABL) && EHX_ABL
    return EHX_OK;  // profiling builds with ablated (wrong-by-construction) kernels: time the first stage only
#endif
    if (n_unc == 0) return EHX_OK;
    HIP_TRY(hipMemsetAsync(d_unc, 0, sizeof(unsigned long long), st));
    std::vector<uint32_t> flags(m);
    HIP_TRY(hipMemcpyAsync(flags.data(), d_flags, m * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    n_short = 0;
    for (size_t j = 0; j < m; ++j)
      if (flags[j]) {
        unc->push_back(subset ? (*subset)[j] : (uint32_t)j);
        n_short += flags[j] == 2u;
      }
    return EHX_OK;
  };

  std::vector<uint32_t> todo, next;
  bool all = true;  // `todo` = every query
  bool counted = false;
  const int eng = resolve_engine(s);
  if (eng == EHX_ENGINE_I8) {
    if (i8_failed) {
      next = *i8_failed;
      n_short = i8_short;
    } else if ((rc = stage(kI8, nullptr, true, &next))) {
      return rc;
    }
    counted = true;
    s->n_i8_queries += nq;
    s->n_i8_fallback += next.size();
    i8_adapt(s, nq, next.size(), n_short, i8_kprime);
    if (next.empty()) return EHX_OK;
    todo.swap(next);
    all = todo.size() * 2 > nq;
  }
  if ((eng == EHX_ENGINE_I8 || eng == EHX_ENGINE_F16) && s->has16 && s->h_unsafe == 0) {
    const size_t m = all ? nq : todo.size();
    if ((rc = stage(kFilter, all ? nullptr : &todo, !counted, &next))) return rc;
    counted = true;
    s->n_filter_queries += m;
    s->n_filter_fallback += next.size();
    if (next.empty()) return EHX_OK;
    todo.swap(next);
    all = todo.size() * 2 > nq;  // most of the batch: just run it all through the fp32 scan
  }
  if ((rc = stage(kF32, all ? nullptr : &todo, !counted, &next))) return rc;
  if (next.empty()) return EHX_OK;
  todo.swap(next);
  // Whatever the matrix-core scans could not certify is answered by the exhaustive canonical pass, kMaxExhaustive
  // queries at a time (bounded scratch): an EHX_OK result is always the certified exhaustive top-k.
  std::vector<uint32_t> chunk;
  for (size_t i0 = 0; i0 < todo.size(); i0 += kMaxExhaustive) {
    chunk.assign(todo.begin() + i0, todo.begin() + std::min(todo.size(), i0 + kMaxExhaustive));
    if ((rc = stage(kExhaustive, &chunk, false, &next))) return rc;
    s->n_exhaustive += chunk.size();
    if (!next.empty()) {  // cannot happen: exact keys are never flagged
      s->n_uncertified_final += next.size();
      return fail(EHX_EINTERNAL, "%zu queries left uncertified by the exhaustive canonical pass", next.size());
    }
  }
  return EHX_OK;
}

}  // namespace ehx_impl
