// ORACLE — TEST INFRASTRUCTURE ONLY (see hnsw_oracle.hpp header).
// extern "C" surface so tests/, smoke() and bench.py's cpu_baseline leg can drive
// the restatement through ctypes.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <string>
#include <thread>
#include <vector>

#include "datagen_oracle.hpp"
#include "hnsw_oracle.hpp"

using namespace oracle;

namespace {
struct Handle {
  size_t dim;
  int metric;
  HnswOracle* hnsw;
};

template <class F>
void parallel_for(size_t n, int threads, F f) {
  if (threads <= 1 || n <= 1) {
    for (size_t i = 0; i < n; i++) f(i, 0);
    return;
  }
  std::atomic<size_t> next(0);
  std::vector<std::thread> pool;
  for (int t = 0; t < threads; t++) {
    pool.emplace_back([&, t]() {
      for (;;) {
        size_t i = next.fetch_add(1);
        if (i >= n) break;
        f(i, t);
      }
    });
  }
  for (auto& th : pool) th.join();
}
}  // namespace

extern "C" {

// ---- raw distance / normalisation ------------------------------------------------
float orc_dist(int metric, const float* a, const float* b, size_t dim) {
  if (metric == METRIC_COSINE) {
    std::vector<float> na(dim), nb(dim);
    normalize_vector(a, na.data(), dim);
    normalize_vector(b, nb.data(), dim);
    return ip_dist(na.data(), nb.data(), dim);
  }
  return metric_dist(metric, a, b, dim);
}
void orc_normalize(const float* in, float* out, size_t dim) { normalize_vector(in, out, dim); }

// ---- HNSW index --------------------------------------------------------------------
void* orc_hnsw_new(size_t dim, int metric, size_t max_elements, size_t M, size_t ef_construction,
                   size_t seed) {
  Handle* h = new Handle;
  h->dim = dim;
  h->metric = metric;
  h->hnsw = new HnswOracle(dim, metric, max_elements, M, ef_construction, seed);
  return h;
}
void orc_hnsw_free(void* p) {
  Handle* h = (Handle*)p;
  delete h->hnsw;
  delete h;
}
// returns 0, or -1 with the exception text in err (cap bytes)
int orc_hnsw_add(void* p, const float* v, uint64_t label, char* err, size_t cap) {
  Handle* h = (Handle*)p;
  try {
    if (h->metric == METRIC_COSINE) {
      std::vector<float> t(h->dim);
      normalize_vector(v, t.data(), h->dim);
      h->hnsw->addPoint(t.data(), label);
    } else {
      h->hnsw->addPoint(v, label);
    }
  } catch (std::exception& e) {
    if (err && cap) snprintf(err, cap, "%s", e.what());
    return -1;
  }
  return 0;
}
// labels = row index; sequential insertion order (what the reference's mutex-serialised
// server produces).  Returns seconds spent.
// CPU BASELINES only: hnswlib's multi-threaded add_items (timing-dependent graph; never used by a parity test).
// Returns seconds spent, or -1 on error (message in err).
double orc_hnsw_add_rows_parallel(void* p, const float* X, size_t n, uint64_t first_label, int threads, char* err,
                                  size_t err_cap) {
  Handle* h = (Handle*)p;
  auto t0 = std::chrono::steady_clock::now();
  try {
    if (h->metric == METRIC_COSINE) {
      std::vector<float> t(n * h->dim);
      for (size_t i = 0; i < n; i++) normalize_vector(X + i * h->dim, t.data() + i * h->dim, h->dim);
      h->hnsw->addPointsParallel(t.data(), n, first_label, threads);
    } else {
      h->hnsw->addPointsParallel(X, n, first_label, threads);
    }
  } catch (const std::exception& e) {
    if (err && err_cap) snprintf(err, err_cap, "%s", e.what());
    return -1.0;
  }
  return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

// STUDIES only: a CPU model of the engine's bulk build (rounds whose rows do not see each other); see hnsw_oracle.hpp.
double orc_hnsw_add_rows_rounds(void* p, const float* X, size_t n, uint64_t first_label, size_t div, size_t cap,
                                int threads, char* err, size_t err_cap) {
  Handle* h = (Handle*)p;
  auto t0 = std::chrono::steady_clock::now();
  try {
    if (h->metric == METRIC_COSINE) {
      std::vector<float> t(n * h->dim);
      for (size_t i = 0; i < n; i++) normalize_vector(X + i * h->dim, t.data() + i * h->dim, h->dim);
      h->hnsw->addPointsRounds(t.data(), n, first_label, div, cap, threads);
    } else {
      h->hnsw->addPointsRounds(X, n, first_label, div, cap, threads);
    }
  } catch (const std::exception& e) {
    if (err && err_cap) snprintf(err, err_cap, "%s", e.what());
    return -1.0;
  }
  return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

double orc_hnsw_add_rows(void* p, const float* X, size_t n, uint64_t first_label) {
  Handle* h = (Handle*)p;
  auto t0 = std::chrono::steady_clock::now();
  std::vector<float> t(h->dim);
  for (size_t i = 0; i < n; i++) {
    const float* v = X + i * h->dim;
    if (h->metric == METRIC_COSINE) {
      normalize_vector(v, t.data(), h->dim);
      v = t.data();
    }
    h->hnsw->addPoint(v, first_label + i);
  }
  return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}
int orc_hnsw_resize(void* p, size_t cap) {
  try {
    ((Handle*)p)->hnsw->resizeIndex(cap);
  } catch (std::exception&) {
    return -1;
  }
  return 0;
}
void orc_hnsw_set_ef(void* p, size_t ef) { ((Handle*)p)->hnsw->setEf(ef); }
size_t orc_hnsw_size(void* p) { return ((Handle*)p)->hnsw->size(); }
int orc_hnsw_maxlevel(void* p) { return ((Handle*)p)->hnsw->maxlevel(); }
uint32_t orc_hnsw_enterpoint(void* p) { return ((Handle*)p)->hnsw->enterpoint(); }
int orc_hnsw_level_of(void* p, uint32_t id) { return ((Handle*)p)->hnsw->level_of(id); }

// nearest-first; returns count (<= k)
size_t orc_hnsw_search(void* p, const float* q, size_t k, uint64_t* labels, float* dists) {
  Handle* h = (Handle*)p;
  std::vector<float> t;
  if (h->metric == METRIC_COSINE) {
    t.resize(h->dim);
    normalize_vector(q, t.data(), h->dim);
    q = t.data();
  }
  auto res = h->hnsw->searchKnn(q, k);
  size_t c = res.size();
  for (size_t i = c; i-- > 0;) {
    dists[i] = res.top().first;
    labels[i] = res.top().second;
    res.pop();
  }
  return c;
}

// Batch search, one query per thread.  stats[5] = n_dist, n_hops0, n_hops_up,
// hnswlib metric_hops, hnswlib metric_distance_computations (totals).  Returns seconds.
double orc_hnsw_search_batch(void* p, const float* Q, size_t nq, size_t k, uint64_t* labels,
                             float* dists, uint32_t* counts, int threads, uint64_t* stats) {
  Handle* h = (Handle*)p;
  if (threads < 1) threads = 1;
  std::vector<SearchCtx> ctx(threads);
  std::vector<std::vector<float>> tmp(threads, std::vector<float>(h->dim));
  auto t0 = std::chrono::steady_clock::now();
  parallel_for(nq, threads, [&](size_t i, int t) {
    const float* q = Q + i * h->dim;
    if (h->metric == METRIC_COSINE) {
      normalize_vector(q, tmp[t].data(), h->dim);
      q = tmp[t].data();
    }
    auto res = h->hnsw->searchKnn(q, k, &ctx[t]);
    size_t c = res.size();
    counts[i] = (uint32_t)c;
    for (size_t j = c; j-- > 0;) {
      dists[i * k + j] = res.top().first;
      labels[i * k + j] = res.top().second;
      res.pop();
    }
  });
  double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (stats) {
    for (int j = 0; j < 5; j++) stats[j] = 0;
    for (auto& c : ctx) {
      stats[0] += c.n_dist;
      stats[1] += c.n_hops0;
      stats[2] += c.n_hops_up;
      stats[3] += c.metric_hops;
      stats[4] += c.metric_distance_computations;
    }
  }
  return sec;
}

// Level-0 search from an explicit entry point with an explicit ef: results sorted
// nearest-first by (dist, id), INTERNAL ids.  For stage-by-stage GPU checks.
size_t orc_hnsw_search_level0(void* p, uint32_t ep, const float* q_prepared, size_t ef,
                              uint32_t* ids, float* dists, uint64_t* stats) {
  Handle* h = (Handle*)p;
  SearchCtx ctx;
  auto top = h->hnsw->searchBaseLayerST(ep, q_prepared, ef, &ctx);
  std::vector<std::pair<float, tableint>> v;
  while (!top.empty()) {
    v.push_back(top.top());
    top.pop();
  }
  std::sort(v.begin(), v.end());
  for (size_t i = 0; i < v.size(); i++) {
    dists[i] = v[i].first;
    ids[i] = v[i].second;
  }
  if (stats) {
    stats[0] = ctx.n_dist;
    stats[1] = ctx.n_hops0;
  }
  return v.size();
}

// Graph export (for the engine's graph-import entry point and for layout checks).
// level0: [n][1+maxM0] u32 (count, ids, zero padded).  levels: [n] int.
void orc_hnsw_export_level0(void* p, uint32_t* out) {
  HnswOracle* g = ((Handle*)p)->hnsw;
  size_t w = g->maxM0() + 1;
  for (size_t i = 0; i < g->size(); i++) {
    const unsigned* ll = g->linklist0((tableint)i);
    out[i * w] = ll[0] & 0xffff;
    for (size_t j = 1; j < w; j++) out[i * w + j] = (j <= (ll[0] & 0xffff)) ? ll[j] : 0;
  }
}
void orc_hnsw_export_levels(void* p, int32_t* out) {
  HnswOracle* g = ((Handle*)p)->hnsw;
  for (size_t i = 0; i < g->size(); i++) out[i] = g->level_of((tableint)i);
}
// upper-level list of node `id` at `level` (>=1): returns count, ids into out[maxM]
uint32_t orc_hnsw_export_upper(void* p, uint32_t id, int level, uint32_t* out) {
  HnswOracle* g = ((Handle*)p)->hnsw;
  const unsigned* ll = g->linklist(id, level);
  uint32_t c = ll[0] & 0xffff;
  for (uint32_t j = 0; j < c; j++) out[j] = ll[1 + j];
  return c;
}
void orc_hnsw_export_vectors(void* p, float* out) {
  Handle* h = (Handle*)p;
  for (size_t i = 0; i < h->hnsw->size(); i++)
    std::memcpy(out + i * h->dim, h->hnsw->vec((tableint)i), sizeof(float) * h->dim);
}

// ---- exhaustive search ---------------------------------------------------------------
// X, Q raw; cosine normalises both (hnswlib-python convention).  Returns seconds of
// the scan itself (normalisation of X excluded).
double orc_exhaustive(const float* X, size_t n, size_t dim, int metric, const float* Q, size_t nq,
                      size_t k, uint64_t* ids, float* dists, uint32_t* counts, int threads) {
  std::vector<float> Xn, Qn;
  const size_t NB = 1024;  // rows per normalisation work item
  if (metric == METRIC_COSINE) {
    Xn.resize(n * dim);
    parallel_for((n + NB - 1) / NB, threads, [&](size_t b, int) {
      const size_t i1 = std::min(n, (b + 1) * NB);
      for (size_t i = b * NB; i < i1; i++) normalize_vector(X + i * dim, &Xn[i * dim], dim);
    });
    X = Xn.data();
    Qn.resize(nq * dim);
    for (size_t i = 0; i < nq; i++) normalize_vector(Q + i * dim, &Qn[i * dim], dim);
    Q = Qn.data();
  }
  auto t0 = std::chrono::steady_clock::now();
  // Row-blocked scan: a thread takes a block of rows small enough to stay in its cache and runs every query over
  // it, keeping the best k of each query it has seen; the per-thread lists are merged at the end.  Same distances
  // (metric_dist) and the same total order (dist, id) as exhaustive_knn over the whole matrix, so the result is
  // identical — only the rows are read from DRAM once instead of once per query.
  typedef std::pair<float, uint64_t> Ent;
  if (threads < 1) threads = 1;
  const size_t RB = std::max<size_t>(16, (size_t)(128 * 1024) / (dim ? dim : 1));
  const size_t nblocks = (n + RB - 1) / RB;
  std::vector<std::vector<std::vector<Ent>>> heaps((size_t)threads);
  parallel_for(nblocks, threads, [&](size_t b, int t) {
    auto& hq = heaps[(size_t)t];
    if (hq.empty()) hq.resize(nq);
    const size_t i0 = b * RB, i1 = std::min(n, i0 + RB);
    for (size_t qi = 0; qi < nq && k > 0; qi++) {
      auto& h = hq[qi];  // max-heap on (dist, id)
      const float* q = Q + qi * dim;
      for (size_t i = i0; i < i1; i++) {
        const Ent e(metric_dist(metric, q, X + i * dim, dim), (uint64_t)i);
        if (h.size() < k) {
          h.push_back(e);
          std::push_heap(h.begin(), h.end());
        } else if (e < h.front()) {
          std::pop_heap(h.begin(), h.end());
          h.back() = e;
          std::push_heap(h.begin(), h.end());
        }
      }
    }
  });
  parallel_for(nq, threads, [&](size_t qi, int) {
    std::vector<Ent> all;
    for (auto& hq : heaps)
      if (!hq.empty()) all.insert(all.end(), hq[qi].begin(), hq[qi].end());
    std::sort(all.begin(), all.end());
    const size_t c = std::min(k, all.size());
    counts[qi] = (uint32_t)c;
    for (size_t j = 0; j < c; j++) {
      dists[qi * k + j] = all[j].first;
      ids[qi * k + j] = all[j].second;
    }
  });
  return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

// ---- ANNIndex wrapper (index.cc) + NearestNeighbor RPC semantics (server.cc:172-210) ---
struct AnnHandle {
  AnnIndexOracle* idx;
  std::unordered_map<std::string, std::vector<float>> store;  // Version::get side copy
  size_t dims;
};
void* orc_ann_new(size_t dims, size_t init_cap, int metric) {
  AnnHandle* h = new AnnHandle;
  h->idx = new AnnIndexOracle(dims, init_cap, metric);
  h->dims = dims;
  return h;
}
void orc_ann_free(void* p) {
  AnnHandle* h = (AnnHandle*)p;
  delete h->idx;
  delete h;
}
void orc_ann_set(void* p, const char* key, const float* v) {
  AnnHandle* h = (AnnHandle*)p;
  h->store[key] = std::vector<float>(v, v + h->dims);
  h->idx->set(key, v);
}
size_t orc_ann_size(void* p) { return ((AnnHandle*)p)->idx->size(); }
static size_t pack_keys(const std::vector<std::string>& keys, char* arena, size_t cap) {
  size_t off = 0;
  for (auto& k : keys) {
    if (off + k.size() + 1 > cap) return (size_t)-1;
    std::memcpy(arena + off, k.data(), k.size());
    arena[off + k.size()] = '\n';
    off += k.size() + 1;
  }
  return off;
}
// keys packed '\n'-separated into arena; returns bytes written (or (size_t)-1)
size_t orc_ann_approx_nearest(void* p, const float* v, size_t num, char* arena, size_t cap,
                              size_t* n_keys) {
  AnnHandle* h = (AnnHandle*)p;
  auto keys = h->idx->approx_nearest(v, num);
  *n_keys = keys.size();
  return pack_keys(keys, arena, cap);
}
// returns gRPC status code (0 OK / 3 INVALID_ARGUMENT / 5 NOT_FOUND)
int orc_ann_nearest_rpc(void* p, int num, const char* key, const float* emb, size_t emb_len,
                        char* arena, size_t cap, size_t* n_keys, size_t* n_bytes) {
  AnnHandle* h = (AnnHandle*)p;
  std::vector<std::string> out;
  int st = nearest_neighbor_rpc(*h->idx, h->store, num, key ? key : "", emb, emb_len, &out);
  *n_keys = out.size();
  *n_bytes = st == 0 ? pack_keys(out, arena, cap) : 0;
  return st;
}

// ---- level generator known-answer helpers ------------------------------
void orc_minstd_first(uint32_t seed, uint32_t* out, size_t n) {
  std::default_random_engine e;
  e.seed(seed);
  for (size_t i = 0; i < n; i++) out[i] = (uint32_t)e();
}
void orc_levels(uint32_t seed, size_t M, int32_t* out, size_t n) {
  std::default_random_engine e;
  e.seed(seed);
  double mult = 1 / log(1.0 * M);
  for (size_t i = 0; i < n; i++) {
    std::uniform_real_distribution<double> distribution(0.0, 1.0);
    out[i] = (int)(-log(distribution(e)) * mult);
  }
}

// ---- synthetic data (EHX-GAUSS-1) ----------------------------------------------------
void orc_philox(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  datagen::U4 c = {{ctr[0], ctr[1], ctr[2], ctr[3]}};
  datagen::U4 r = datagen::philox4x32_10(c, key[0], key[1]);
  for (int i = 0; i < 4; i++) out[i] = r.v[i];
}
void orc_gen_rows(uint64_t seed, uint64_t row0, size_t n_rows, size_t dim, int normalize, float* out,
                  int threads) {
  parallel_for(n_rows, threads,
               [&](size_t i, int) { datagen::gen_row(seed, row0 + i, dim, normalize != 0, out + i * dim); });
}

void orc_gen_manifold_rows(uint64_t seed, uint64_t row0, size_t n_rows, size_t dim, uint32_t latent, int normalize,
                           float* out, int threads) {
  if (latent == 0 || latent > 64) return;
  std::vector<float> basis((size_t)latent * dim);
  datagen::manifold_basis(dim, latent, basis.data());
  parallel_for(n_rows, threads, [&](size_t i, int) {
    datagen::gen_manifold_row(seed, row0 + i, dim, latent, basis.data(), normalize != 0, out + i * dim);
  });
}

}  // extern "C"
