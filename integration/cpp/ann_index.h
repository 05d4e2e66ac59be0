// Drop-in replacement for embeddinghub/embeddingstore/index.h:19-33 (class ANNIndex) on the MI355X
// engine: same class name, namespace, constructor and method signatures, so embeddingstore's
// version.cc / server.cc compile unchanged against it; the hnswlib members are replaced by an
// engine space reached through the C ABI (include/ehx.h).
#pragma once

#include <atomic>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "ehx.h"

namespace featureform {
namespace embedding {

class ANNIndex {
 public:
  // index.cc:10-18.  `init_cap` keeps its meaning (rows reserved up front; doubles when full).
  explicit ANNIndex(size_t dims, size_t init_cap = 128) : dims_(dims), space_(nullptr) {
    static std::atomic<unsigned long> counter{0};
    name_ = "annindex-" + std::to_string(counter++);
    ehx_params p{};
    p.mode = EHX_MODE_FLAT;          // exact neighbours; EHX_MODE_GRAPH for HNSW-style search
    p.initial_capacity = init_cap;
    check(ehx_space_create(name_.data(), name_.size(), (uint32_t)dims, EHX_METRIC_L2SQ /* index.cc:13 */,
                           EHX_DTYPE_F32, &p, &space_));
  }
  ~ANNIndex() {
    if (space_) ehx_space_drop(space_);
  }
  ANNIndex(const ANNIndex&) = delete;
  ANNIndex& operator=(const ANNIndex&) = delete;

  // index.cc:20-37
  void set(std::string key, std::vector<float> value) {
    if (value.size() != dims_) throw std::invalid_argument("ANNIndex::set: wrong dimension");
    check(ehx_set(space_, key.data(), key.size(), value.data()));
  }

  // index.cc:39-52 — nearest first.  Returns fewer than `num` keys only when the index holds fewer
  // than `num` vectors (the reference has undefined behaviour there).
  std::vector<std::string> approx_nearest(std::vector<float> value, size_t num) const {
    std::vector<std::string> keys;
    if (num == 0) return keys;
    if (value.size() != dims_) throw std::invalid_argument("ANNIndex::approx_nearest: wrong dimension");
    std::vector<uint64_t> ids(num), off(num + 1);
    std::vector<float> dist(num);
    uint32_t count = 0;
    std::vector<char> arena(4096);
    for (;;) {
      int rc = ehx_knn_keys(space_, 1, value.data(), (uint32_t)num, ids.data(), dist.data(), &count, arena.data(),
                            arena.size(), off.data());
      if (rc == EHX_ERANGE) {
        arena.resize(arena.size() * 4);
        continue;
      }
      check(rc);
      break;
    }
    for (uint32_t j = 0; j < count; ++j) keys.emplace_back(arena.data() + off[j], arena.data() + off[j + 1]);
    return keys;
  }

 private:
  static void check(int rc) {
    if (rc != EHX_OK) throw std::runtime_error(std::string("ehx: ") + ehx_last_error());
  }
  size_t dims_;
  std::string name_;
  ehx_space* space_;
};

}  // namespace embedding
}  // namespace featureform
