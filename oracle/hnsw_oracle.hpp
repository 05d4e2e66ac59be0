// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// CPU restatement of the nearest-neighbour path that featureform/embeddinghub
// delegates to hnswlib.  Nothing under embeddinghub_amd/ (the product) may
// include, link or call this file; only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg use it, and only as the checker / CPU baseline.
//
// Provenance.  The arithmetic of the path is NOT in /root/reference: it lives in
// the third-party dependency nmslib/hnswlib pinned at git commit
// 21b54fe9544cfbb757b2ea8f3def5542ba2435c7 (embeddinghub/WORKSPACE:80-85) and in
// PyPI hnswlib==0.5.2 (embeddinghub/sdk/python/requirements.txt:1).  Neither is
// vendored or installable here (no network), so this file restates hnswlib's
// published algorithm (hnswalg.h / space_l2.h / space_ip.h of the 0.5.x line) from
// its documented behaviour, and parity is anchored on the reference's own call
// sites and known-answer tests:
//   * embeddinghub/embeddingstore/index.cc:10-18   ctor: L2Space(dims), all HNSW defaults
//   * embeddinghub/embeddingstore/index.cc:20-37   set(): dense labels, capacity doubling, addPoint upsert
//   * embeddinghub/embeddingstore/index.cc:39-52   approx_nearest(): searchKnn + reverse
//   * embeddinghub/embeddingstore/server.cc:172-210 NearestNeighbor RPC semantics
//   * embeddinghub/embeddingstore/test/index_test.cc:17-60   4 known-answer tests
//   * embeddinghub/sdk/python/offlinehub.py:27-141 + test/offlinehub_test.py:63-86
// PINNING STATUS: pinned against every known-answer test the reference holds for
// this path (tests/test_oracle_golden.py); hnswlib itself could not be run here, so
// bit-level agreement with an hnswlib binary on large random inputs is unpinned.
//
// Arithmetic model: the reference is built by Bazel with default x86-64 flags, i.e.
// hnswlib's SSE code path with separate (non-fused) multiply and add.  This file is
// compiled with -ffp-contract=off so that `a*b + c` is never contracted into an FMA.
#pragma once

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <memory>
#include <mutex>
#include <queue>
#include <random>
#include <stdexcept>
#include <string>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <utility>
#include <vector>

#include <emmintrin.h>  // SSE2: the reference's build uses hnswlib's USE_SSE path

namespace oracle {

typedef unsigned int tableint;
typedef size_t labeltype;

enum Metric { METRIC_L2 = 0, METRIC_IP = 1, METRIC_COSINE = 2 };

// ---------------------------------------------------------------------------
// Distance functions (hnswlib space_l2.h / space_ip.h, SSE variants).
// Dispatch rule (L2Space / InnerProductSpace constructors):
//   dim % 16 == 0 -> SIMD16Ext ; dim % 4 == 0 -> SIMD4Ext ;
//   dim > 16 -> SIMD16ExtResiduals ; dim > 4 -> SIMD4ExtResiduals ; else scalar.
// ---------------------------------------------------------------------------
static inline float l2_scalar(const float* a, const float* b, size_t qty) {
  float res = 0;
  for (size_t i = 0; i < qty; i++) {
    float t = a[i] - b[i];
    res += t * t;
  }
  return res;
}

static inline float hsum4_ordered(__m128 v) {
  float t[4];
  _mm_storeu_ps(t, v);
  return t[0] + t[1] + t[2] + t[3];  // left-to-right, as hnswlib's TmpRes sum
}

static inline float l2_simd16(const float* a, const float* b, size_t qty) {
  size_t qty16 = qty >> 4;
  const float* end = a + (qty16 << 4);
  __m128 sum = _mm_set1_ps(0);
  while (a < end) {
    for (int u = 0; u < 4; u++) {
      __m128 v1 = _mm_loadu_ps(a);
      a += 4;
      __m128 v2 = _mm_loadu_ps(b);
      b += 4;
      __m128 diff = _mm_sub_ps(v1, v2);
      sum = _mm_add_ps(sum, _mm_mul_ps(diff, diff));
    }
  }
  return hsum4_ordered(sum);
}

static inline float l2_simd4(const float* a, const float* b, size_t qty) {
  size_t qty4 = qty >> 2;
  const float* end = a + (qty4 << 2);
  __m128 sum = _mm_set1_ps(0);
  while (a < end) {
    __m128 v1 = _mm_loadu_ps(a);
    a += 4;
    __m128 v2 = _mm_loadu_ps(b);
    b += 4;
    __m128 diff = _mm_sub_ps(v1, v2);
    sum = _mm_add_ps(sum, _mm_mul_ps(diff, diff));
  }
  return hsum4_ordered(sum);
}

static inline float l2_dist(const float* a, const float* b, size_t dim) {
  if (dim % 16 == 0) return l2_simd16(a, b, dim);
  if (dim % 4 == 0) return l2_simd4(a, b, dim);
  if (dim > 16) {
    size_t q16 = dim >> 4 << 4;
    float res = l2_simd16(a, b, q16);
    float tail = l2_scalar(a + q16, b + q16, dim - q16);
    return res + tail;
  }
  if (dim > 4) {
    size_t q4 = dim >> 2 << 2;
    float res = l2_simd4(a, b, q4);
    float tail = l2_scalar(a + q4, b + q4, dim - q4);
    return res + tail;
  }
  return l2_scalar(a, b, dim);
}

static inline float ip_scalar_sum(const float* a, const float* b, size_t qty) {
  float res = 0;
  for (size_t i = 0; i < qty; i++) res += a[i] * b[i];
  return res;
}

static inline float ip_simd16_sum(const float* a, const float* b, size_t qty) {
  size_t qty16 = qty / 16;
  const float* end = a + 16 * qty16;
  __m128 sum = _mm_set1_ps(0);
  while (a < end) {
    for (int u = 0; u < 4; u++) {
      __m128 v1 = _mm_loadu_ps(a);
      a += 4;
      __m128 v2 = _mm_loadu_ps(b);
      b += 4;
      sum = _mm_add_ps(sum, _mm_mul_ps(v1, v2));
    }
  }
  return hsum4_ordered(sum);
}

// hnswlib's InnerProductSIMD4Ext (SSE): 16-wide body first, then 4-wide steps.
static inline float ip_simd4_sum(const float* a, const float* b, size_t qty) {
  size_t qty16 = qty / 16;
  size_t qty4 = qty / 4;
  const float* end1 = a + 16 * qty16;
  const float* end2 = a + 4 * qty4;
  __m128 sum = _mm_set1_ps(0);
  while (a < end1) {
    for (int u = 0; u < 4; u++) {
      __m128 v1 = _mm_loadu_ps(a);
      a += 4;
      __m128 v2 = _mm_loadu_ps(b);
      b += 4;
      sum = _mm_add_ps(sum, _mm_mul_ps(v1, v2));
    }
  }
  while (a < end2) {
    __m128 v1 = _mm_loadu_ps(a);
    a += 4;
    __m128 v2 = _mm_loadu_ps(b);
    b += 4;
    sum = _mm_add_ps(sum, _mm_mul_ps(v1, v2));
  }
  return hsum4_ordered(sum);
}

static inline float ip_dist(const float* a, const float* b, size_t dim) {
  if (dim % 16 == 0) return 1.0f - ip_simd16_sum(a, b, dim);
  if (dim % 4 == 0) return 1.0f - ip_simd4_sum(a, b, dim);
  // The residual variants of hnswlib 0.5.x (space_ip.h: InnerProductSIMD16ExtResiduals / SIMD4ExtResiduals) call two
  // functions that each already return a DISTANCE, 1 - sum, and combine them as  res + res_tail - 1.0f  — three
  // roundings, not the two of 1 - (sum + tail).  [upstream-recall; VERDICT r01 flagged the earlier form]
  if (dim > 16) {
    size_t q16 = dim >> 4 << 4;
    float res = 1.0f - ip_simd16_sum(a, b, q16);
    float tail = 1.0f - ip_scalar_sum(a + q16, b + q16, dim - q16);
    return res + tail - 1.0f;
  }
  if (dim > 4) {
    size_t q4 = dim >> 2 << 2;
    float res = 1.0f - ip_simd4_sum(a, b, q4);
    float tail = 1.0f - ip_scalar_sum(a + q4, b + q4, dim - q4);
    return res + tail - 1.0f;
  }
  return 1.0f - ip_scalar_sum(a, b, dim);
}

// hnswlib python bindings, Index::normalize_vector (cosine space = normalise on
// add and on query, then InnerProductSpace).
static inline void normalize_vector(const float* data, float* out, size_t dim) {
  float norm = 0.0f;
  for (size_t i = 0; i < dim; i++) norm += data[i] * data[i];
  norm = 1.0f / (sqrtf(norm) + 1e-30f);
  for (size_t i = 0; i < dim; i++) out[i] = data[i] * norm;
}

// Distance between two STORED/PREPARED vectors (cosine inputs already normalised).
static inline float metric_dist(int metric, const float* a, const float* b, size_t dim) {
  return metric == METRIC_L2 ? l2_dist(a, b, dim) : ip_dist(a, b, dim);
}

// ---------------------------------------------------------------------------
// HierarchicalNSW restatement.
// ---------------------------------------------------------------------------
struct CompareByFirst {
  constexpr bool operator()(std::pair<float, tableint> const& a,
                            std::pair<float, tableint> const& b) const noexcept {
    return a.first < b.first;
  }
};
typedef std::priority_queue<std::pair<float, tableint>, std::vector<std::pair<float, tableint>>,
                            CompareByFirst>
    CandQueue;

// Per-caller search state: visited tags + work counters.  The index owns one
// (used by the single-threaded build path); concurrent searchers bring their own.
struct alignas(128) SearchCtx {  // own cache lines: contexts of concurrent searchers sit in one vector
  std::vector<unsigned short> visited;
  unsigned short tag = 0;
  // SURVEY §8d work counters: n_dist = distances actually evaluated,
  // n_hops0 / n_hops_up = expanded nodes at level 0 / upper levels; metric_* are
  // hnswlib's own looser counters (metric_distance_computations adds whole list sizes).
  uint64_t n_dist = 0, n_hops0 = 0, n_hops_up = 0;
  uint64_t metric_hops = 0, metric_distance_computations = 0;
  void reset_counters() { n_dist = n_hops0 = n_hops_up = metric_hops = metric_distance_computations = 0; }
  void next_tag(size_t n) {
    if (visited.size() < n) { visited.assign(n, 0); tag = 0; }
    tag++;
    if (tag == 0) { std::fill(visited.begin(), visited.end(), 0); tag = 1; }
  }
};

class HnswOracle {
 public:
  // Defaults as used by the reference (index.cc:14-15 passes only space+capacity):
  // M=16, ef_construction=200, random_seed=100, ef=10.
  HnswOracle(size_t dim, int metric, size_t max_elements, size_t M = 16,
             size_t ef_construction = 200, size_t random_seed = 100)
      : dim_(dim), metric_(metric), max_elements_(max_elements) {
    M_ = M;
    maxM_ = M_;
    maxM0_ = M_ * 2;
    ef_construction_ = std::max(ef_construction, M_);
    ef_ = 10;
    level_generator_.seed(random_seed);
    update_probability_generator_.seed(random_seed + 1);
    mult_ = 1 / log(1.0 * M_);
    cur_element_count_ = 0;
    enterpoint_node_ = (tableint)-1;
    maxlevel_ = -1;
    data_.resize(max_elements_ * dim_);
    link0_.assign(max_elements_ * (maxM0_ + 1), 0);
    links_upper_.resize(max_elements_);
    element_levels_.assign(max_elements_, 0);
  }

  void setEf(size_t ef) { ef_ = ef; }
  size_t ef() const { return ef_; }
  size_t size() const { return cur_element_count_; }
  size_t capacity() const { return max_elements_; }
  size_t dim() const { return dim_; }
  int maxlevel() const { return maxlevel_; }
  tableint enterpoint() const { return enterpoint_node_; }
  size_t M() const { return M_; }
  size_t maxM0() const { return maxM0_; }
  int level_of(tableint id) const { return element_levels_[id]; }
  const float* vec(tableint id) const { return &data_[(size_t)id * dim_]; }

  const unsigned* linklist0(tableint id) const { return &link0_[(size_t)id * (maxM0_ + 1)]; }
  unsigned* linklist0(tableint id) { return &link0_[(size_t)id * (maxM0_ + 1)]; }
  const unsigned* linklist(tableint id, int level) const {
    return &links_upper_[id][(size_t)(level - 1) * (maxM_ + 1)];
  }
  unsigned* linklist(tableint id, int level) {
    return &links_upper_[id][(size_t)(level - 1) * (maxM_ + 1)];
  }
  const unsigned* linklist_at(tableint id, int level) const {
    return level == 0 ? linklist0(id) : linklist(id, level);
  }
  unsigned* linklist_at(tableint id, int level) {
    return level == 0 ? linklist0(id) : linklist(id, level);
  }

  void resizeIndex(size_t new_max) {
    if (new_max < cur_element_count_)
      throw std::runtime_error("Cannot resize, max element is less than the current number of elements");
    data_.resize(new_max * dim_);
    link0_.resize(new_max * (maxM0_ + 1), 0);
    links_upper_.resize(new_max);
    element_levels_.resize(new_max, 0);
    max_elements_ = new_max;
  }

  float dist(const float* a, const float* b) const { return metric_dist(metric_, a, b, dim_); }

  int getRandomLevel(double reverse_size) {
    std::uniform_real_distribution<double> distribution(0.0, 1.0);
    double r = -log(distribution(level_generator_)) * reverse_size;
    return (int)r;
  }

  // addPoint(data, label): insert, or update-in-place when the label exists.
  // `data` must already be prepared for the metric (normalised for cosine).
  tableint addPoint(const float* data_point, labeltype label) {
    auto search = label_lookup_.find(label);
    if (search != label_lookup_.end()) {
      tableint existing = search->second;
      updatePoint(data_point, existing, 1.0f);
      return existing;
    }
    if (cur_element_count_ >= max_elements_)
      throw std::runtime_error("The number of elements exceeds the specified limit");
    tableint cur_c = (tableint)cur_element_count_;
    cur_element_count_++;
    label_lookup_[label] = cur_c;
    if (labels_.size() <= cur_c) labels_.resize(cur_c + 1);
    labels_[cur_c] = label;

    int curlevel = getRandomLevel(mult_);
    element_levels_[cur_c] = curlevel;
    int maxlevelcopy = maxlevel_;
    tableint currObj = enterpoint_node_;

    std::memset(linklist0(cur_c), 0, sizeof(unsigned) * (maxM0_ + 1));
    std::memcpy(&data_[(size_t)cur_c * dim_], data_point, sizeof(float) * dim_);
    links_upper_[cur_c].assign((size_t)curlevel * (maxM_ + 1), 0);

    if (currObj != (tableint)-1) {
      if (curlevel < maxlevelcopy) {
        float curdist = dist(data_point, vec(currObj));
        for (int level = maxlevelcopy; level > curlevel; level--) {
          bool changed = true;
          while (changed) {
            changed = false;
            const unsigned* data = linklist(currObj, level);
            int size = (int)(data[0] & 0xffff);
            const tableint* datal = data + 1;
            for (int i = 0; i < size; i++) {
              tableint cand = datal[i];
              float d = dist(data_point, vec(cand));
              if (d < curdist) {
                curdist = d;
                currObj = cand;
                changed = true;
              }
            }
          }
        }
      }
      for (int level = std::min(curlevel, maxlevelcopy); level >= 0; level--) {
        CandQueue top_candidates = searchBaseLayer(currObj, data_point, level);
        currObj = mutuallyConnectNewElement(data_point, cur_c, top_candidates, level, false);
      }
    } else {
      enterpoint_node_ = 0;
      maxlevel_ = curlevel;
    }
    if (curlevel > maxlevelcopy) {
      enterpoint_node_ = cur_c;
      maxlevel_ = curlevel;
    }
    return cur_c;
  }

  // searchKnn(q, k): result as a max-heap of (dist, label), like hnswlib.
  std::priority_queue<std::pair<float, labeltype>> searchKnn(const float* query, size_t k,
                                                             SearchCtx* pctx = nullptr) const {
    SearchCtx& ctx = pctx ? *pctx : ctx0_;
    std::priority_queue<std::pair<float, labeltype>> result;
    if (cur_element_count_ == 0) return result;
    tableint currObj = enterpoint_node_;
    float curdist = dist(query, vec(enterpoint_node_));
    ctx.n_dist++;
    for (int level = maxlevel_; level > 0; level--) {
      bool changed = true;
      while (changed) {
        changed = false;
        const unsigned* data = linklist(currObj, level);
        int size = (int)(data[0] & 0xffff);
        ctx.metric_hops++;
        ctx.n_hops_up++;
        ctx.metric_distance_computations += size;
        const tableint* datal = data + 1;
        for (int i = 0; i < size; i++) {
          tableint cand = datal[i];
          float d = dist(query, vec(cand));
          ctx.n_dist++;
          if (d < curdist) {
            curdist = d;
            currObj = cand;
            changed = true;
          }
        }
      }
    }
    CandQueue top_candidates = searchBaseLayerST(currObj, query, std::max(ef_, k), &ctx);
    while (top_candidates.size() > k) top_candidates.pop();
    while (top_candidates.size() > 0) {
      std::pair<float, tableint> rez = top_candidates.top();
      result.push(std::pair<float, labeltype>(rez.first, labels_[rez.second]));
      top_candidates.pop();
    }
    return result;
  }

  // Level-0 best-first search starting at a given entry point (exposed so the
  // GPU graph kernel can be checked stage by stage).
  CandQueue searchBaseLayerST(tableint ep_id, const float* data_point, size_t ef,
                              SearchCtx* pctx = nullptr) const {
    SearchCtx& ctx = pctx ? *pctx : ctx0_;
    ctx.next_tag(max_elements_);
    std::vector<unsigned short>& visited_ = ctx.visited;
    const unsigned short visited_tag_ = ctx.tag;
    CandQueue top_candidates;
    CandQueue candidate_set;
    float lowerBound;
    {
      float d = dist(data_point, vec(ep_id));
      ctx.n_dist++;
      lowerBound = d;
      top_candidates.emplace(d, ep_id);
      candidate_set.emplace(-d, ep_id);
    }
    visited_[ep_id] = visited_tag_;
    while (!candidate_set.empty()) {
      std::pair<float, tableint> current_node_pair = candidate_set.top();
      if ((-current_node_pair.first) > lowerBound) break;
      candidate_set.pop();
      tableint current_node_id = current_node_pair.second;
      const unsigned* data = linklist0(current_node_id);
      size_t size = data[0] & 0xffff;
      ctx.metric_hops++;
      ctx.n_hops0++;
      ctx.metric_distance_computations += size;
      for (size_t j = 1; j <= size; j++) {
        tableint candidate_id = data[j];
        if (!(visited_[candidate_id] == visited_tag_)) {
          visited_[candidate_id] = visited_tag_;
          float d = dist(data_point, vec(candidate_id));
          ctx.n_dist++;
          if (top_candidates.size() < ef || lowerBound > d) {
            candidate_set.emplace(-d, candidate_id);
            top_candidates.emplace(d, candidate_id);
            if (top_candidates.size() > ef) top_candidates.pop();
            if (!top_candidates.empty()) lowerBound = top_candidates.top().first;
          }
        }
      }
    }
    return top_candidates;
  }

  SearchCtx& ctx() const { return ctx0_; }

  labeltype label_of(tableint id) const { return labels_[id]; }

 private:
  CandQueue searchBaseLayer(tableint ep_id, const float* data_point, int layer) {
    ctx0_.next_tag(max_elements_);
    std::vector<unsigned short>& visited_ = ctx0_.visited;
    const unsigned short visited_tag_ = ctx0_.tag;
    CandQueue top_candidates;
    CandQueue candidateSet;
    float lowerBound;
    {
      float d = dist(data_point, vec(ep_id));
      top_candidates.emplace(d, ep_id);
      lowerBound = d;
      candidateSet.emplace(-d, ep_id);
    }
    visited_[ep_id] = visited_tag_;
    while (!candidateSet.empty()) {
      std::pair<float, tableint> curr_el_pair = candidateSet.top();
      if ((-curr_el_pair.first) > lowerBound) break;
      candidateSet.pop();
      tableint curNodeNum = curr_el_pair.second;
      const unsigned* data = linklist_at(curNodeNum, layer);
      size_t size = data[0] & 0xffff;
      const tableint* datal = data + 1;
      for (size_t j = 0; j < size; j++) {
        tableint candidate_id = datal[j];
        if (visited_[candidate_id] == visited_tag_) continue;
        visited_[candidate_id] = visited_tag_;
        float dist1 = dist(data_point, vec(candidate_id));
        if (top_candidates.size() < ef_construction_ || lowerBound > dist1) {
          candidateSet.emplace(-dist1, candidate_id);
          top_candidates.emplace(dist1, candidate_id);
          if (top_candidates.size() > ef_construction_) top_candidates.pop();
          if (!top_candidates.empty()) lowerBound = top_candidates.top().first;
        }
      }
    }
    return top_candidates;
  }

  void getNeighborsByHeuristic2(CandQueue& top_candidates, const size_t M) {
    if (top_candidates.size() < M) return;
    std::priority_queue<std::pair<float, tableint>> queue_closest;
    std::vector<std::pair<float, tableint>> return_list;
    while (top_candidates.size() > 0) {
      queue_closest.emplace(-top_candidates.top().first, top_candidates.top().second);
      top_candidates.pop();
    }
    while (queue_closest.size()) {
      if (return_list.size() >= M) break;
      std::pair<float, tableint> curent_pair = queue_closest.top();
      float dist_to_query = -curent_pair.first;
      queue_closest.pop();
      bool good = true;
      for (std::pair<float, tableint> second_pair : return_list) {
        float curdist = dist(vec(second_pair.second), vec(curent_pair.second));
        if (curdist < dist_to_query) {
          good = false;
          break;
        }
      }
      if (good) return_list.push_back(curent_pair);
    }
    for (std::pair<float, tableint> curent_pair : return_list)
      top_candidates.emplace(-curent_pair.first, curent_pair.second);
  }

  tableint mutuallyConnectNewElement(const float* /*data_point*/, tableint cur_c,
                                     CandQueue& top_candidates, int level, bool isUpdate) {
    size_t Mcurmax = level ? maxM_ : maxM0_;
    getNeighborsByHeuristic2(top_candidates, M_);
    if (top_candidates.size() > M_)
      throw std::runtime_error("Should be not be more than M_ candidates returned by the heuristic");
    std::vector<tableint> selectedNeighbors;
    selectedNeighbors.reserve(M_);
    while (top_candidates.size() > 0) {
      selectedNeighbors.push_back(top_candidates.top().second);
      top_candidates.pop();
    }
    tableint next_closest_entry_point = selectedNeighbors.back();
    {
      unsigned* ll_cur = linklist_at(cur_c, level);
      if ((ll_cur[0] & 0xffff) && !isUpdate)
        throw std::runtime_error("The newly inserted element should have blank link list");
      ll_cur[0] = (unsigned)selectedNeighbors.size();
      tableint* data = ll_cur + 1;
      for (size_t idx = 0; idx < selectedNeighbors.size(); idx++) {
        if (level > element_levels_[selectedNeighbors[idx]])
          throw std::runtime_error("Trying to make a link on a non-existent level");
        data[idx] = selectedNeighbors[idx];
      }
    }
    for (size_t idx = 0; idx < selectedNeighbors.size(); idx++) {
      unsigned* ll_other = linklist_at(selectedNeighbors[idx], level);
      size_t sz_link_list_other = ll_other[0] & 0xffff;
      if (sz_link_list_other > Mcurmax) throw std::runtime_error("Bad value of sz_link_list_other");
      if (selectedNeighbors[idx] == cur_c)
        throw std::runtime_error("Trying to connect an element to itself");
      if (level > element_levels_[selectedNeighbors[idx]])
        throw std::runtime_error("Trying to make a link on a non-existent level");
      tableint* data = ll_other + 1;
      bool is_cur_c_present = false;
      if (isUpdate) {
        for (size_t j = 0; j < sz_link_list_other; j++) {
          if (data[j] == cur_c) {
            is_cur_c_present = true;
            break;
          }
        }
      }
      if (!is_cur_c_present) {
        if (sz_link_list_other < Mcurmax) {
          data[sz_link_list_other] = cur_c;
          ll_other[0] = (unsigned)(sz_link_list_other + 1);
        } else {
          float d_max = dist(vec(cur_c), vec(selectedNeighbors[idx]));
          CandQueue candidates;
          candidates.emplace(d_max, cur_c);
          for (size_t j = 0; j < sz_link_list_other; j++)
            candidates.emplace(dist(vec(data[j]), vec(selectedNeighbors[idx])), data[j]);
          getNeighborsByHeuristic2(candidates, Mcurmax);
          int indx = 0;
          while (candidates.size() > 0) {
            data[indx] = candidates.top().second;
            candidates.pop();
            indx++;
          }
          ll_other[0] = (unsigned)indx;
        }
      }
    }
    return next_closest_entry_point;
  }

  std::vector<tableint> getConnections(tableint id, int level) const {
    const unsigned* data = linklist_at(id, level);
    int size = (int)(data[0] & 0xffff);
    std::vector<tableint> result(size);
    std::memcpy(result.data(), data + 1, size * sizeof(tableint));
    return result;
  }

  void updatePoint(const float* dataPoint, tableint internalId, float updateNeighborProbability) {
    std::memcpy(&data_[(size_t)internalId * dim_], dataPoint, sizeof(float) * dim_);
    int maxLevelCopy = maxlevel_;
    tableint entryPointCopy = enterpoint_node_;
    if (entryPointCopy == internalId && cur_element_count_ == 1) return;
    int elemLevel = element_levels_[internalId];
    std::uniform_real_distribution<float> distribution(0.0, 1.0);
    for (int layer = 0; layer <= elemLevel; layer++) {
      std::unordered_set<tableint> sCand;
      std::unordered_set<tableint> sNeigh;
      std::vector<tableint> listOneHop = getConnections(internalId, layer);
      if (listOneHop.size() == 0) continue;
      sCand.insert(internalId);
      for (auto&& elOneHop : listOneHop) {
        sCand.insert(elOneHop);
        if (distribution(update_probability_generator_) > updateNeighborProbability) continue;
        sNeigh.insert(elOneHop);
        std::vector<tableint> listTwoHop = getConnections(elOneHop, layer);
        for (auto&& elTwoHop : listTwoHop) sCand.insert(elTwoHop);
      }
      for (auto&& neigh : sNeigh) {
        CandQueue candidates;
        size_t size = sCand.find(neigh) == sCand.end() ? sCand.size() : sCand.size() - 1;
        size_t elementsToKeep = std::min(ef_construction_, size);
        for (auto&& cand : sCand) {
          if (cand == neigh) continue;
          float distance = dist(vec(neigh), vec(cand));
          if (candidates.size() < elementsToKeep) {
            candidates.emplace(distance, cand);
          } else {
            if (distance < candidates.top().first) {
              candidates.pop();
              candidates.emplace(distance, cand);
            }
          }
        }
        getNeighborsByHeuristic2(candidates, layer == 0 ? maxM0_ : maxM_);
        {
          unsigned* ll_cur = linklist_at(neigh, layer);
          size_t candSize = candidates.size();
          ll_cur[0] = (unsigned)candSize;
          tableint* data = ll_cur + 1;
          for (size_t idx = 0; idx < candSize; idx++) {
            data[idx] = candidates.top().second;
            candidates.pop();
          }
        }
      }
    }
    repairConnectionsForUpdate(dataPoint, entryPointCopy, internalId, elemLevel, maxLevelCopy);
  }

  void repairConnectionsForUpdate(const float* dataPoint, tableint entryPointInternalId,
                                  tableint dataPointInternalId, int dataPointLevel, int maxLevel) {
    tableint currObj = entryPointInternalId;
    if (dataPointLevel < maxLevel) {
      float curdist = dist(dataPoint, vec(currObj));
      for (int level = maxLevel; level > dataPointLevel; level--) {
        bool changed = true;
        while (changed) {
          changed = false;
          const unsigned* data = linklist_at(currObj, level);
          int size = (int)(data[0] & 0xffff);
          const tableint* datal = data + 1;
          for (int i = 0; i < size; i++) {
            tableint cand = datal[i];
            float d = dist(dataPoint, vec(cand));
            if (d < curdist) {
              curdist = d;
              currObj = cand;
              changed = true;
            }
          }
        }
      }
    }
    if (dataPointLevel > maxLevel)
      throw std::runtime_error("Level of item to be updated cannot be bigger than max level");
    for (int level = dataPointLevel; level >= 0; level--) {
      CandQueue topCandidates = searchBaseLayer(currObj, dataPoint, level);
      CandQueue filteredTopCandidates;
      while (topCandidates.size() > 0) {
        if (topCandidates.top().second != dataPointInternalId)
          filteredTopCandidates.push(topCandidates.top());
        topCandidates.pop();
      }
      if (filteredTopCandidates.size() > 0) {
        currObj = mutuallyConnectNewElement(dataPoint, dataPointInternalId, filteredTopCandidates,
                                            level, true);
      }
    }
  }

  // -----------------------------------------------------------------------------------------------
  // Parallel bulk build — hnswlib's multi-threaded add_items (hnswlib-python; the reference's offline path calls it at
  // sdk/python/offlinehub.py:89 with the default thread count).  ONLY for CPU BASELINES of bench.py and studies: like
  // hnswlib's, the graph it builds depends on thread timing, so no parity test ever uses it — the sequential addPoint
  // above is the oracle, and nothing below is called by it or changes it (separate functions on purpose).
  // Locking as in hnswlib: one mutex per element guards its link lists (held while a searcher reads a list and while
  // mutuallyConnect rewrites it), the element being inserted holds its own lock throughout, and an insertion that
  // raises the top level holds the global lock until it has become the entry point.  Rows are fresh labels only; the
  // first row of an empty index is inserted alone (hnswlib does the same); levels are drawn up front, in row order.
  // -----------------------------------------------------------------------------------------------
 public:
  void addPointsParallel(const float* rows, size_t n, labeltype first_label, int threads) {
    if (n == 0) return;
    if (cur_element_count_ + n > max_elements_)
      throw std::runtime_error("The number of elements exceeds the specified limit");
    for (size_t i = 0; i < n; ++i)
      if (label_lookup_.count(first_label + i)) throw std::runtime_error("parallel build: labels must be fresh");
    size_t start = 0;
    if (cur_element_count_ == 0) {  // the first element alone
      addPoint(rows, first_label);
      start = 1;
    }
    const size_t base = cur_element_count_;
    const size_t m = n - start;
    if (m == 0) return;
    if (labels_.size() < base + m) labels_.resize(base + m);
    for (size_t i = 0; i < m; ++i) {  // ids, levels, vectors, empty lists: before any thread starts
      const tableint id = (tableint)(base + i);
      label_lookup_[first_label + start + i] = id;
      labels_[id] = first_label + start + i;
      const int lvl = getRandomLevel(mult_);
      element_levels_[id] = lvl;
      std::memset(linklist0(id), 0, sizeof(unsigned) * (maxM0_ + 1));
      std::memcpy(&data_[(size_t)id * dim_], rows + (start + i) * dim_, sizeof(float) * dim_);
      links_upper_[id].assign((size_t)lvl * (maxM_ + 1), 0);
    }
    cur_element_count_ = base + m;
    if (mt_locks_.size() < max_elements_) mt_locks_ = std::vector<std::mutex>(max_elements_);
    if (threads < 1) threads = 1;
    std::atomic<size_t> next{0};
    std::vector<std::thread> pool;
    std::vector<std::string> errors((size_t)threads);
    for (int t = 0; t < threads; ++t)
      pool.emplace_back([&, t] {
        SearchCtx ctx;
        try {
          for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= m) break;
            insertLinkedMT((tableint)(base + i), ctx);
          }
        } catch (const std::exception& e) {
          errors[(size_t)t] = e.what();
          next.store(m);
        }
      });
    for (auto& th : pool) th.join();
    for (auto& e : errors)
      if (!e.empty()) throw std::runtime_error("parallel build: " + e);
  }

 private:
  void insertLinkedMT(tableint cur_c, SearchCtx& ctx) {
    const float* data_point = vec(cur_c);
    const int curlevel = element_levels_[cur_c];
    std::unique_lock<std::mutex> lock_el(mt_locks_[cur_c]);
    std::unique_lock<std::mutex> templock(mt_global_);
    const int maxlevelcopy = maxlevel_;
    tableint currObj = enterpoint_node_;
    if (curlevel <= maxlevelcopy) templock.unlock();
    if (curlevel < maxlevelcopy) {
      float curdist = dist(data_point, vec(currObj));
      for (int level = maxlevelcopy; level > curlevel; level--) {
        bool changed = true;
        while (changed) {
          changed = false;
          std::unique_lock<std::mutex> lk(mt_locks_[currObj]);
          const unsigned* data = linklist(currObj, level);
          const int size = (int)(data[0] & 0xffff);
          const tableint* datal = data + 1;
          for (int i = 0; i < size; i++) {
            const tableint cand = datal[i];
            const float d = dist(data_point, vec(cand));
            if (d < curdist) {
              curdist = d;
              currObj = cand;
              changed = true;
            }
          }
        }
      }
    }
    for (int level = std::min(curlevel, maxlevelcopy); level >= 0; level--) {
      CandQueue top = searchBaseLayerMT(currObj, data_point, level, ctx);
      currObj = connectMT(cur_c, top, level);
    }
    if (curlevel > maxlevelcopy) {  // (still under the global lock)
      enterpoint_node_ = cur_c;
      maxlevel_ = curlevel;
    }
  }

  CandQueue searchBaseLayerMT(tableint ep_id, const float* data_point, int layer, SearchCtx& ctx) {
    ctx.next_tag(max_elements_);
    std::vector<unsigned short>& visited = ctx.visited;
    const unsigned short tag = ctx.tag;
    CandQueue top_candidates, candidateSet;
    float lowerBound;
    {
      const float d = dist(data_point, vec(ep_id));
      top_candidates.emplace(d, ep_id);
      lowerBound = d;
      candidateSet.emplace(-d, ep_id);
    }
    visited[ep_id] = tag;
    std::vector<tableint> nbrs;
    while (!candidateSet.empty()) {
      const std::pair<float, tableint> curr = candidateSet.top();
      if ((-curr.first) > lowerBound) break;
      candidateSet.pop();
      {
        std::unique_lock<std::mutex> lk(mt_locks_[curr.second]);  // the list may be rewritten by a concurrent insertion
        const unsigned* data = linklist_at(curr.second, layer);
        const size_t size = data[0] & 0xffff;
        nbrs.assign(data + 1, data + 1 + size);
      }
      for (const tableint cand : nbrs) {
        if (visited[cand] == tag) continue;
        visited[cand] = tag;
        const float d1 = dist(data_point, vec(cand));
        if (top_candidates.size() < ef_construction_ || lowerBound > d1) {
          candidateSet.emplace(-d1, cand);
          top_candidates.emplace(d1, cand);
          if (top_candidates.size() > ef_construction_) top_candidates.pop();
          if (!top_candidates.empty()) lowerBound = top_candidates.top().first;
        }
      }
    }
    return top_candidates;
  }

  // mutuallyConnectNewElement with the neighbour's lock held while its list changes (cur_c's own lock is held by the
  // caller); a link may only go to an element that has the level (elements of the same batch may still be unlinked
  // at it — their lists are valid, empty, and fill when their own insertion gets there)
  tableint connectMT(tableint cur_c, CandQueue& top_candidates, int level) {
    const size_t Mcurmax = level ? maxM_ : maxM0_;
    getNeighborsByHeuristic2(top_candidates, M_);
    std::vector<tableint> selected;
    selected.reserve(M_);
    while (top_candidates.size() > 0) {
      selected.push_back(top_candidates.top().second);
      top_candidates.pop();
    }
    const tableint next_closest_entry_point = selected.back();
    {
      unsigned* ll_cur = linklist_at(cur_c, level);
      ll_cur[0] = (unsigned)selected.size();
      for (size_t idx = 0; idx < selected.size(); idx++) ll_cur[1 + idx] = selected[idx];
    }
    for (const tableint other : selected) {
      std::unique_lock<std::mutex> lk(mt_locks_[other]);
      unsigned* ll_other = linklist_at(other, level);
      const size_t sz = ll_other[0] & 0xffff;
      tableint* data = ll_other + 1;
      if (sz < Mcurmax) {
        data[sz] = cur_c;
        ll_other[0] = (unsigned)(sz + 1);
      } else {
        CandQueue candidates;
        candidates.emplace(dist(vec(cur_c), vec(other)), cur_c);
        for (size_t j = 0; j < sz; j++) candidates.emplace(dist(vec(data[j]), vec(other)), data[j]);
        getNeighborsByHeuristic2(candidates, Mcurmax);
        int indx = 0;
        while (candidates.size() > 0) {
          data[indx++] = candidates.top().second;
          candidates.pop();
        }
        ll_other[0] = (unsigned)indx;
      }
    }
    return next_closest_entry_point;
  }

  // -----------------------------------------------------------------------------------------------
  // A CPU MODEL of the engine's BULK build (embeddinghub_amd/csrc/ehx_api.cpp graph_insert + k_insert.hip) — for
  // studies only (scripts/studies/bulk_build_model.py): rows join the graph in ROUNDS; every row of a round runs
  // hnswlib's insertion search on the graph as it was BEFORE the round (so the rows of a round do not see each other),
  // selects its neighbours with the heuristic, and then the round is linked in ascending id order: a new node's own
  // lists are written as selected, every selected neighbour gets the back-link (appended, or its list re-selected
  // with the heuristic when it is full) — mutuallyConnectNewElement with the candidates of the frozen graph.  Round
  // size: graph_size / div, at most cap, 1 while the graph holds fewer than 64 nodes (graph_insert's round_size).
  // Levels come from the same generator in the same order as the sequential build's.
  // -----------------------------------------------------------------------------------------------
 public:
  void addPointsRounds(const float* rows, size_t n, labeltype first_label, size_t div, size_t cap, int threads) {
    if (cur_element_count_ + n > max_elements_)
      throw std::runtime_error("The number of elements exceeds the specified limit");
    size_t pos = 0;
    if (cur_element_count_ == 0 && n > 0) {
      addPoint(rows, first_label);
      pos = 1;
    }
    struct Plan {
      std::vector<std::vector<tableint>> sel;  // [level] selected neighbours, farthest first (the heap's pop order)
    };
    if (threads < 1) threads = 1;
    while (pos < n) {
      const size_t g = cur_element_count_;
      size_t P = 1;
      if (g >= 64) P = std::max<size_t>(1, std::min(cap, g / div));
      P = std::min(P, n - pos);
      const size_t base = cur_element_count_;
      if (labels_.size() < base + P) labels_.resize(base + P);
      for (size_t i = 0; i < P; ++i) {  // register the round's rows (levels in row order); lists stay empty for now
        const tableint id = (tableint)(base + i);
        label_lookup_[first_label + pos + i] = id;
        labels_[id] = first_label + pos + i;
        const int lvl = getRandomLevel(mult_);
        element_levels_[id] = lvl;
        std::memset(linklist0(id), 0, sizeof(unsigned) * (maxM0_ + 1));
        std::memcpy(&data_[(size_t)id * dim_], rows + (pos + i) * dim_, sizeof(float) * dim_);
        links_upper_[id].assign((size_t)lvl * (maxM_ + 1), 0);
      }
      // phase 1: searches on the frozen graph (cur_element_count_ still excludes the round: nothing links to it)
      std::vector<Plan> plans(P);
      const int maxlevelcopy = maxlevel_;
      const tableint ep = enterpoint_node_;
      std::atomic<size_t> next{0};
      auto work = [&] {
        SearchCtx ctx;
        for (;;) {
          const size_t i = next.fetch_add(1);
          if (i >= P) break;
          const tableint cur_c = (tableint)(base + i);
          const float* q = vec(cur_c);
          const int curlevel = element_levels_[cur_c];
          tableint currObj = ep;
          if (curlevel < maxlevelcopy) {
            float curdist = dist(q, vec(currObj));
            for (int level = maxlevelcopy; level > curlevel; level--) {
              bool changed = true;
              while (changed) {
                changed = false;
                const unsigned* data = linklist(currObj, level);
                const int size = (int)(data[0] & 0xffff);
                for (int j = 0; j < size; j++) {
                  const tableint cand = data[1 + j];
                  const float dd = dist(q, vec(cand));
                  if (dd < curdist) {
                    curdist = dd;
                    currObj = cand;
                    changed = true;
                  }
                }
              }
            }
          }
          const int top = std::min(curlevel, maxlevelcopy);
          plans[i].sel.assign((size_t)top + 1, {});
          for (int level = top; level >= 0; level--) {
            CandQueue cand = searchBaseLayerMT(currObj, q, level, ctx);  // (locks uncontended: nothing writes now)
            getNeighborsByHeuristic2(cand, M_);
            std::vector<tableint>& sel = plans[i].sel[(size_t)level];
            while (cand.size() > 0) {
              sel.push_back(cand.top().second);
              cand.pop();
            }
            currObj = sel.back();
          }
        }
      };
      if (mt_locks_.size() < max_elements_) mt_locks_ = std::vector<std::mutex>(max_elements_);
      std::vector<std::thread> pool;
      for (int t = 1; t < threads && (size_t)t < P; ++t) pool.emplace_back(work);
      work();
      for (auto& th : pool) th.join();
      // phase 2: link the round in ascending id order
      for (size_t i = 0; i < P; ++i) {
        const tableint cur_c = (tableint)(base + i);
        for (size_t level = 0; level < plans[i].sel.size(); ++level) {
          const std::vector<tableint>& sel = plans[i].sel[level];
          const size_t Mcurmax = level ? maxM_ : maxM0_;
          unsigned* ll_cur = linklist_at(cur_c, (int)level);
          ll_cur[0] = (unsigned)sel.size();
          for (size_t x = 0; x < sel.size(); ++x) ll_cur[1 + x] = sel[x];
          for (const tableint other : sel) {
            unsigned* ll_other = linklist_at(other, (int)level);
            const size_t sz = ll_other[0] & 0xffff;
            tableint* data = ll_other + 1;
            if (sz < Mcurmax) {
              data[sz] = cur_c;
              ll_other[0] = (unsigned)(sz + 1);
            } else {
              CandQueue candidates;
              candidates.emplace(dist(vec(cur_c), vec(other)), cur_c);
              for (size_t j = 0; j < sz; j++) candidates.emplace(dist(vec(data[j]), vec(other)), data[j]);
              getNeighborsByHeuristic2(candidates, Mcurmax);
              int indx = 0;
              while (candidates.size() > 0) {
                data[indx++] = candidates.top().second;
                candidates.pop();
              }
              ll_other[0] = (unsigned)indx;
            }
          }
        }
      }
      cur_element_count_ = base + P;
      for (size_t i = 0; i < P; ++i) {  // entry point / top level: the first node of the round that raises it
        const tableint id = (tableint)(base + i);
        if (element_levels_[id] > maxlevel_) {
          enterpoint_node_ = id;
          maxlevel_ = element_levels_[id];
        }
      }
      pos += P;
    }
  }

 private:
  std::vector<std::mutex> mt_locks_;
  std::mutex mt_global_;

  size_t dim_;
  int metric_;
  size_t max_elements_;
  size_t cur_element_count_;
  size_t M_, maxM_, maxM0_, ef_construction_, ef_;
  double mult_;
  int maxlevel_;
  tableint enterpoint_node_;
  std::vector<float> data_;
  std::vector<unsigned> link0_;                    // [max][1 + maxM0]: count, ids
  std::vector<std::vector<unsigned>> links_upper_;  // per element: level x (1 + maxM)
  std::vector<int> element_levels_;
  std::vector<labeltype> labels_;
  std::unordered_map<labeltype, tableint> label_lookup_;
  mutable SearchCtx ctx0_;
  std::default_random_engine level_generator_;
  std::default_random_engine update_probability_generator_;
};

// ---------------------------------------------------------------------------
// Exhaustive search: ground truth for recall and the CPU comparator of the
// brute-force path.  Same distance arithmetic as above; ordered by (dist, id).
// ---------------------------------------------------------------------------
static inline void exhaustive_knn(const float* X, size_t n, size_t dim, int metric, const float* q,
                                  size_t k, uint64_t* out_ids, float* out_dist, size_t* out_count) {
  std::priority_queue<std::pair<float, uint64_t>> heap;  // max-heap on (dist, id)
  for (size_t i = 0; i < n; i++) {
    float d = metric_dist(metric, q, X + i * dim, dim);
    if (heap.size() < k) {
      heap.emplace(d, (uint64_t)i);
    } else if (k > 0 && std::make_pair(d, (uint64_t)i) < heap.top()) {
      heap.pop();
      heap.emplace(d, (uint64_t)i);
    }
  }
  size_t c = heap.size();
  *out_count = c;
  for (size_t i = c; i-- > 0;) {
    out_dist[i] = heap.top().first;
    out_ids[i] = heap.top().second;
    heap.pop();
  }
}

// ---------------------------------------------------------------------------
// ANNIndex wrapper semantics — embeddinghub/embeddingstore/index.{h,cc}.
// ---------------------------------------------------------------------------
class AnnIndexOracle {
 public:
  // index.cc:10-18 (L2Space, HNSW defaults, init_cap 128 per index.h:21).  `metric`
  // is an extension knob; the reference is METRIC_L2.
  explicit AnnIndexOracle(size_t dims, size_t init_cap = 128, int metric = METRIC_L2)
      : dims_(dims), capacity_(init_cap), metric_(metric),
        nn_(new HnswOracle(dims, metric, init_cap)), next_label_(0) {}

  // index.cc:20-37
  void set(const std::string& key, const float* value) {
    auto it = key_to_label_.find(key);
    labeltype label;
    if (it == key_to_label_.end()) {
      label = next_label_;
      next_label_++;
      label_to_key_[label] = key;
      key_to_label_[key] = label;
      if (next_label_ == capacity_) {
        capacity_ *= 2;
        nn_->resizeIndex(capacity_);
      }
    } else {
      label = it->second;
    }
    if (metric_ == METRIC_COSINE) {
      std::vector<float> tmp(dims_);
      normalize_vector(value, tmp.data(), dims_);
      nn_->addPoint(tmp.data(), label);
    } else {
      nn_->addPoint(value, label);
    }
  }

  // index.cc:39-52.  The reference has UB when the index holds fewer than `num`
  // elements; the oracle returns what exists (documented divergence).
  std::vector<std::string> approx_nearest(const float* value, size_t num) const {
    std::vector<float> tmp;
    const float* q = value;
    if (metric_ == METRIC_COSINE) {
      tmp.resize(dims_);
      normalize_vector(value, tmp.data(), dims_);
      q = tmp.data();
    }
    auto pairs = nn_->searchKnn(q, num);
    std::vector<std::string> keys(pairs.size());
    for (int i = (int)pairs.size() - 1; i >= 0; i--) {
      keys[i] = label_to_key_.at(pairs.top().second);
      pairs.pop();
    }
    return keys;
  }

  bool has_key(const std::string& key) const { return key_to_label_.count(key) != 0; }
  size_t size() const { return next_label_; }
  HnswOracle* hnsw() { return nn_.get(); }
  const HnswOracle* hnsw() const { return nn_.get(); }
  bool label_of(const std::string& key, labeltype* out) const {
    auto it = key_to_label_.find(key);
    if (it == key_to_label_.end()) return false;
    *out = it->second;
    return true;
  }

 private:
  size_t dims_;
  size_t capacity_;
  int metric_;
  std::unique_ptr<HnswOracle> nn_;
  std::unordered_map<std::string, labeltype> key_to_label_;
  std::unordered_map<labeltype, std::string> label_to_key_;
  labeltype next_label_;
};

// server.cc:172-210 — NearestNeighbor RPC semantics over an index + a key->vector
// store.  Status: 0 OK, 3 INVALID_ARGUMENT, 5 NOT_FOUND (gRPC codes).  The NOT_FOUND
// for an unknown key replaces the reference's UB.
static inline int nearest_neighbor_rpc(const AnnIndexOracle& idx,
                                       const std::unordered_map<std::string, std::vector<float>>& store,
                                       int num, const std::string& key, const float* embedding,
                                       size_t embedding_len, std::vector<std::string>* out) {
  bool has_key = key != "";
  bool has_vec = embedding_len != 0;
  if (has_key && has_vec) return 3;
  if (!has_key && !has_vec) return 3;
  std::vector<float> ref_vec;
  size_t num_retrieve = (size_t)num;
  if (has_key) {
    auto it = store.find(key);
    if (it == store.end()) return 5;
    ref_vec = it->second;
    num_retrieve += 1;
  } else {
    ref_vec.assign(embedding, embedding + embedding_len);
  }
  std::vector<std::string> nearest = idx.approx_nearest(ref_vec.data(), num_retrieve);
  if (has_key) {
    auto it = std::find(nearest.begin(), nearest.end(), key);
    if (it != nearest.end()) {
      nearest.erase(it);
    } else if (!nearest.empty()) {
      nearest.pop_back();
    }
  }
  *out = nearest;
  return 0;
}

}  // namespace oracle
