// The four known-answer tests of embeddinghub/embeddingstore/test/index_test.cc:17-60, unchanged in
// substance, run against the drop-in ANNIndex (integration/cpp/ann_index.h) on the GPU engine.
// gtest is not in this image, so plain asserts; build: see tests/test_cpp_dropin.py.
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "ann_index.h"

using featureform::embedding::ANNIndex;

#define EXPECT_EQ_KEYS(got, ...)                                                            \
  do {                                                                                      \
    std::vector<std::string> expected{__VA_ARGS__};                                         \
    if ((got) != expected) {                                                                \
      std::fprintf(stderr, "FAILED %s:%d\n", __FILE__, __LINE__);                           \
      for (auto& k : (got)) std::fprintf(stderr, "  got '%s'\n", k.c_str());                \
      std::exit(1);                                                                         \
    }                                                                                       \
  } while (0)

static std::unique_ptr<ANNIndex> make() {
  auto idx = std::make_unique<ANNIndex>(3);
  idx->set("a", std::vector<float>{0, 1, 0});
  idx->set("b", std::vector<float>{1, 1, 0});
  idx->set("c", std::vector<float>{1, 0, 0});
  return idx;
}

int main() {
  auto a_vec = std::vector<float>{0, 1, 0};
  {  // TestSimpleANN
    auto idx = make();
    EXPECT_EQ_KEYS(idx->approx_nearest(a_vec, 1), "a");
  }
  {  // TestMultiANN
    auto idx = make();
    EXPECT_EQ_KEYS(idx->approx_nearest(a_vec, 2), "a", "b");
  }
  {  // TestUpdateANN
    auto idx = make();
    idx->set("a", std::vector<float>{0, -1, 0});
    EXPECT_EQ_KEYS(idx->approx_nearest(a_vec, 1), "b");
  }
  {  // TestANN0Items
    auto idx = make();
    auto got = idx->approx_nearest(a_vec, 0);
    if (!got.empty()) return 1;
  }
  std::puts("index_test: 4 passed");
  return 0;
}
