// Internal launch interface between the C-ABI host code (ehx_api.cpp) and the gfx950 kernels.
// Not part of the public boundary (include/ehx.h is).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ehx {

constexpr uint32_t kTileRows = 128;   // corpus rows per scan tile
constexpr uint32_t kTileQ = 256;      // queries per scan tile
constexpr uint32_t kBK = 32;          // k-depth of one LDS stage (floats)
constexpr uint32_t kCandSlots = 64;   // per-(query, block) candidate slots = one wave row
constexpr uint64_t kKeyInf = ~0ull;

// (score, id) packed so that unsigned 64-bit order == (score asc, id asc).
__host__ __device__ inline uint32_t f32_to_ordered(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
  uint32_t u = __float_as_uint(f);
#else
  uint32_t u;
  __builtin_memcpy(&u, &f, 4);
#endif
  return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__host__ __device__ inline float ordered_to_f32(uint32_t o) {
  uint32_t u = o ^ ((o >> 31) ? 0x80000000u : 0xFFFFFFFFu);
#if defined(__HIP_DEVICE_COMPILE__)
  return __uint_as_float(u);
#else
  float f;
  __builtin_memcpy(&f, &u, 4);
  return f;
#endif
}

// Exactly-rounded single operations for the canonical (oracle-order) arithmetic.  HIP's
// __fmul_rn/__fadd_rn are plain * and + (contractible) and __fsqrt_rn is the approximate native
// sqrt, so they are NOT used; the library is built with -ffp-contract=off and relies on hipcc's
// default -fhip-fp32-correctly-rounded-divide-sqrt for / and sqrt.
__device__ __forceinline__ float ex_add(float a, float b) { return a + b; }
__device__ __forceinline__ float ex_sub(float a, float b) { return a - b; }
__device__ __forceinline__ float ex_mul(float a, float b) { return a * b; }
__device__ __forceinline__ float ex_div(float a, float b) { return a / b; }
__device__ __forceinline__ float ex_sqrt(float a) { return __builtin_sqrtf(a); }

struct ScanArgs {
  const float* Q;        // [q_tiles*256][ld] prepared queries (zero padded)
  const float* X;        // [cap][ld] stored rows, cap % 256 == 0 (>= n_tiles*128), pad columns zero
  const float2* rowp;    // [cap] epilogue (a, b): approx distance = dot*a + b
  uint64_t* cand;        // [grid][256][64] per-block candidate slots (scratch)
  uint64_t* part;        // [q_tiles*256][n_chunks][kprime] sorted partial top-k' keys
  uint32_t n;            // valid rows
  uint32_t ld;           // row stride in floats, % 32 == 0
  uint32_t n_tiles;      // ceil(n / 128)
  uint32_t q_tiles;
  uint32_t n_chunks;
  uint32_t tiles_per_chunk;
  uint32_t kprime;       // <= 64
  uint32_t* err;         // device error counter (bounded-retry guard tripped)
  uint32_t xcd_map;      // 1: blocks of one chunk share an XCD (grid % 8 == 0, n_chunks % 8 == 0)
};

size_t scan_lds_bytes();
hipError_t launch_flat_scan(const ScanArgs& a, hipStream_t st);

// one wave per query: k-way merge of the per-chunk sorted key lists -> top-kprime keys
hipError_t launch_flat_merge(const uint64_t* part, uint32_t nq, uint32_t n_chunks, uint32_t kprime,
                             uint64_t* merged /*[nq][64]*/, hipStream_t st);

// canonical (oracle-order) distances of the merged candidates, sort by (dist, id), emit top-k.
struct RerankArgs {
  const float* Q;          // prepared queries [*][ld]
  const float* X;
  const float* inv_norm;   // [cap] (cosine) or nullptr
  const uint64_t* merged;  // [nq][64] keys (approx score, id)
  uint64_t* out_ids;       // [nq][k]
  float* out_dist;         // [nq][k]
  uint32_t* out_count;     // [nq]
  unsigned long long* n_uncertified;  // device counter
  uint32_t nq, k, kprime, n, dims, ld;
  int metric;
};
hipError_t launch_rerank(const RerankArgs& a, hipStream_t st);

// prepared queries: copy into the padded [q_rows][ld] buffer, L2-normalise for cosine
hipError_t launch_prep_queries(const float* q_in, uint32_t nq, uint32_t dims, uint32_t ld,
                               uint32_t q_rows, int metric, float* q_out, hipStream_t st);

// per-row statistics for rows [row0, row0+n): inv_norm (cosine), rowp (a,b) for the scan epilogue
hipError_t launch_row_stats(const float* X, uint64_t row0, uint64_t n, uint32_t dims, uint32_t ld,
                            int metric, float* inv_norm, float2* rowp, hipStream_t st);
// rowp for padding rows [row0, row0+n): (0, +inf)
hipError_t launch_rowp_pad(float2* rowp, uint64_t row0, uint64_t n, hipStream_t st);

// EHX-GAUSS-1 rows generated straight into a [*, ld] matrix (optionally L2-normalised)
hipError_t launch_gen_rows(uint64_t seed, uint64_t row0, uint64_t n_rows, uint32_t dims, uint32_t ld,
                           int normalize, float* out, hipStream_t st);

// k-way merge of per-shard (dist, id) result lists [n_lists][nq][k] -> [nq][k]
hipError_t launch_merge_lists(const uint64_t* ids, const float* dist, const uint32_t* count,
                              uint32_t nq, uint32_t k, uint32_t n_lists, uint64_t* out_ids,
                              float* out_dist, uint32_t* out_count, hipStream_t st);

}  // namespace ehx
