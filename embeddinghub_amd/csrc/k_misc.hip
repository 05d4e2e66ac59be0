// Small kernels around the scan: query preparation, per-row statistics maintained at Set time,
// and the EHX-GAUSS-1 synthetic workload generator (include/ehx_datagen.h).
//
// Canonical arithmetic: the L2 norm used for cosine follows hnswlib-python's
// Index::normalize_vector (sequential fp32 sum, non-fused; norm = 1/(sqrt(sum)+1e-30f); x*norm) —
// the convention SURVEY.md §8d fixes and oracle/hnsw_oracle.hpp:normalize_vector restates.
#include "../../include/ehx_datagen.h"
#include "ehx_kernels.h"
#include "k_prep_query.h"
#include <map>
#include <mutex>

namespace ehx {

// work-items per dispatch (a grid dimension holds fewer than 2^32): launches that scale with the row count go in chunks
constexpr uint64_t kMaxWorkItems = 1ull << 31;

namespace {
// one thread walks one row sequentially: the summation ORDER is the contract here
template <typename XT>
__device__ __forceinline__ float seq_sumsq(const XT* __restrict__ x, uint32_t dims) {
  float s = 0.0f;
  uint32_t i = 0;
  // eight loads in flight ahead of the (strictly sequential) adds: the ORDER of the sum is unchanged
  for (; i + 8 <= dims; i += 8) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = ld_row(x, i + j);
#pragma unroll
    for (int j = 0; j < 8; ++j) s = ex_add(s, ex_mul(v[j], v[j]));
  }
  for (; i < dims; ++i) {
    const float v = ld_row(x, i);
    s = ex_add(s, ex_mul(v, v));
  }
  return s;
}
}  // namespace

// (inv_norm_of, prep_query_row: k_prep_query.h — the graph search kernel prepares a single query itself in its one-launch form)

__global__ __launch_bounds__(64) void prep_queries_kernel(const float* __restrict__ q_in, uint32_t nq,
                                                          uint32_t dims, uint32_t ld, uint32_t q_rows,
                                                          int metric, float* __restrict__ q_out) {
  prep_query_row(q_in, nq, dims, ld, metric, q_out, blockIdx.x, (int)threadIdx.x);
}

hipError_t launch_prep_queries(const float* q_in, uint32_t nq, uint32_t dims, uint32_t ld, uint32_t q_rows,
                               int metric, float* q_out, hipStream_t st) {
  hipLaunchKernelGGL(prep_queries_kernel, dim3(q_rows), dim3(64), 0, st, q_in, nq, dims, ld, q_rows,
                     metric, q_out);
  return hipGetLastError();
}

// the same sequential sum over a row stored block-permuted (single-copy graph spaces: element m sits at
// search_copy_pos(m)) — the ORDER of the additions is the logical one
__device__ __forceinline__ float seq_sumsq_perm(const float* __restrict__ x, uint32_t dims) {
  float s = 0.0f;
  uint32_t i = 0;
  for (; i + 8 <= dims; i += 8) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = x[search_copy_pos(i + j)];
#pragma unroll
    for (int j = 0; j < 8; ++j) s = ex_add(s, ex_mul(v[j], v[j]));
  }
  for (; i < dims; ++i) {
    const float v = x[search_copy_pos(i)];
    s = ex_add(s, ex_mul(v, v));
  }
  return s;
}

template <typename XT>
__global__ __launch_bounds__(256) void row_stats_kernel(const XT* __restrict__ X, uint64_t row0,
                                                        uint64_t n, uint32_t dims, uint32_t ld, int metric,
                                                        float* __restrict__ inv_norm,
                                                        float2* __restrict__ rowp,
                                                        unsigned int* __restrict__ max_sumsq, int perm) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t r = row0 + i;
  float s;
  if constexpr (sizeof(XT) == 4) s = perm ? seq_sumsq_perm((const float*)(X + r * ld), dims) : seq_sumsq(X + r * ld, dims);
  else s = seq_sumsq(X + r * ld, dims);
  // largest |x|^2 ever written to the space (bit pattern of a non-negative float orders like the float;
  // NaN / Inf rows park it at +Inf): the re-rank's certification margin needs a bound of every row's norm
  if (max_sumsq) atomicMax(max_sumsq, (s == s) ? __float_as_uint(s) : 0x7F800000u);
  if (metric == 2) {
    const float inv = inv_norm_of(s);
    inv_norm[r] = inv;
    rowp[r] = make_float2(-inv, 1.0f);  // approx cosine distance = 1 - dot(q^, x) * inv
  } else if (metric == 0) {
    rowp[r] = make_float2(-2.0f, s);  // approx L2^2 - |q|^2 = |x|^2 - 2 dot
  } else {
    rowp[r] = make_float2(-1.0f, 1.0f);  // 1 - dot
  }
}

hipError_t launch_row_stats(const void* X, int x_half, uint64_t row0, uint64_t n, uint32_t dims, uint32_t ld,
                            int metric, float* inv_norm, float2* rowp, float* max_sumsq, hipStream_t st, int perm) {
  if (n == 0) return hipSuccess;
  const uint32_t grid = (uint32_t)((n + 255) / 256);
  if (x_half)
    hipLaunchKernelGGL(row_stats_kernel<__half>, dim3(grid), dim3(256), 0, st, (const __half*)X, row0, n, dims, ld,
                       metric, inv_norm, rowp, (unsigned int*)max_sumsq, 0);
  else
    hipLaunchKernelGGL(row_stats_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)X, row0, n, dims, ld,
                       metric, inv_norm, rowp, (unsigned int*)max_sumsq, perm);
  return hipGetLastError();
}

// ---- single-copy graph spaces: the rows are stored ONCE, in the search copy's block order, raw -------------------
// Within every 16-float block element 4 i + j moves to 4 j + i (search_copy_pos: a 4 x 4 transpose, its own inverse).
// In place: one thread per block.  ids (optional): the rows to permute (rows written in place); else rows
// [row0, row0 + n).  Applied exactly once to a freshly written row — a second application would undo it.
__global__ __launch_bounds__(256) void permute_blocks_kernel(float* __restrict__ X, uint32_t ld, uint64_t row0, uint64_t n,
                                                             const uint64_t* __restrict__ ids) {
  const uint64_t e = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  const uint32_t bpr = ld >> 4;   // 16-float blocks per row (ld % 32 == 0)
  if (e >= n * bpr) return;
  const uint64_t i = e / bpr;
  const uint64_t r = ids ? ids[i] : row0 + i;
  float4* b = (float4*)(X + r * ld + (size_t)(e % bpr) * 16);
  const float4 v0 = b[0], v1 = b[1], v2 = b[2], v3 = b[3];
  b[0] = make_float4(v0.x, v1.x, v2.x, v3.x);
  b[1] = make_float4(v0.y, v1.y, v2.y, v3.y);
  b[2] = make_float4(v0.z, v1.z, v2.z, v3.z);
  b[3] = make_float4(v0.w, v1.w, v2.w, v3.w);
}

hipError_t launch_permute_blocks(float* X, uint32_t ld, uint64_t row0, uint64_t n, const uint64_t* ids, hipStream_t st) {
  if (n == 0) return hipSuccess;
  const uint64_t max_rows = kMaxWorkItems / (ld >> 4);
  for (uint64_t r0 = 0; r0 < n; r0 += max_rows) {
    const uint64_t m = n - r0 < max_rows ? n - r0 : max_rows;
    hipLaunchKernelGGL(permute_blocks_kernel, dim3((uint32_t)((m * (ld >> 4) + 255) / 256)), dim3(256), 0, st, X, ld,
                       row0 + r0, m, ids ? ids + r0 : nullptr);
  }
  return hipGetLastError();
}

// ---- graph mode: the search copy of the rows (k_graph.hip) ------------------------------------------
// Xs[row][16t + 4j + i] = X[row][16t + 4i + j] (* inv_norm[row] for cosine: hnswlib-python stores the
// normalised row, one rounding per element — the same product the distance kernels otherwise form on
// the fly).  Within every 16-float block the four inputs of SSE partial sum j become 16 contiguous
// bytes, so a 4-lane group reads a block as ONE coalesced 64-byte piece and lane j gets exactly its
// partial sum's inputs, in order.  Pad columns stay zero.
namespace {
template <typename XT>
__global__ __launch_bounds__(256) void make_search_copy_kernel(const XT* __restrict__ X, const float* __restrict__ inv_norm,
                                                               uint64_t row0, uint64_t n, uint32_t ld, int scale,
                                                               float* __restrict__ Xs) {
  const uint64_t e = (uint64_t)blockIdx.x * 256 + threadIdx.x;  // element index within the row range
  if (e >= n * ld) return;
  const uint64_t r = row0 + e / ld;
  const uint32_t c = (uint32_t)(e % ld);                        // destination column
  const uint32_t blk = c & ~15u, j = (c >> 2) & 3u, i = c & 3u;
  float v = (float)X[r * ld + blk + 4 * i + j];  // (binary16 rows: the rounded value, exactly)
  if (scale) v = ex_mul(v, inv_norm[r]);
  Xs[r * ld + c] = v;
}
}  // namespace

hipError_t launch_make_search_copy(const void* X, bool x_half, const float* inv_norm, uint64_t row0, uint64_t n,
                                   uint32_t ld, int metric, float* Xs, hipStream_t st) {
  if (n == 0) return hipSuccess;
  // A dispatch holds fewer than 2^32 work-items per grid dimension: one thread per element of a 10 M x 768 fill is
  // 7.7 * 10^9 — launched in one go (round 2 did) only the first (n * ld) mod 2^32 elements, 4.4 M rows, were written
  // and the rest of the search copy was whatever the allocation held.  Row chunks of at most 2^31 elements each.
  const uint64_t max_rows = kMaxWorkItems / ld;
  for (uint64_t r0 = 0; r0 < n; r0 += max_rows) {
    const uint64_t m = n - r0 < max_rows ? n - r0 : max_rows;
    const dim3 grid((uint32_t)((m * ld + 255) / 256));
    if (x_half)
      hipLaunchKernelGGL(make_search_copy_kernel<_Float16>, grid, dim3(256), 0, st, (const _Float16*)X, inv_norm,
                         row0 + r0, m, ld, metric == 2 ? 1 : 0, Xs);
    else
      hipLaunchKernelGGL(make_search_copy_kernel<float>, grid, dim3(256), 0, st, (const float*)X, inv_norm, row0 + r0, m,
                         ld, metric == 2 ? 1 : 0, Xs);
  }
  return hipGetLastError();
}

// ---- fp16-MFMA filter scan: scan copy and query preparation (k_flat16.hip) ---------------------------
// One wave per row.  The norm here is the filter's own (parallel fp32 sum; its rounding is inside the
// eps of scan16_eps) — the canonical distances never see it.
namespace {
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// norms the filter's error analysis covers: finite, and far from the fp32 underflow/overflow ranges
__device__ __forceinline__ bool norm_ok(float sumsq) { return sumsq == 0.0f || (sumsq > 1e-24f && sumsq < 1e30f); }
}  // namespace

template <typename XT>
__global__ __launch_bounds__(256) void make_scan16_kernel(const XT* __restrict__ X, uint64_t row0, uint64_t n,
                                                          uint32_t dims, uint32_t ld, uint32_t ld16, int metric,
                                                          __half* __restrict__ X16, float2* __restrict__ rowp16,
                                                          unsigned long long* __restrict__ n_unsafe) {
  const int lane = threadIdx.x & 63;
  const uint64_t i = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  const uint64_t r = row0 + i;
  const XT* x = X + r * ld;
  float ss = 0.0f;
  for (uint32_t c = lane; c < dims; c += 64) ss += ld_row(x, c) * ld_row(x, c);
  ss = wave_sum(ss);
  const bool ok = norm_ok(ss);
  const float nr = ok ? __builtin_sqrtf(ss) : 0.0f;
  const float inv = nr > 0.0f ? 1.0f / nr : 0.0f;
  for (uint32_t c = lane; c < ld16; c += 64) X16[scan16_index(r, c, ld16)] = __float2half_rn(c < dims ? ld_row(x, c) * inv : 0.0f);
  if (lane == 0) {
    float2 p;
    if (!ok) {
      p = make_float2(0.0f, __builtin_inff());
      atomicAdd(n_unsafe, 1ull);
    } else if (metric == 2) {
      p = make_float2(-1.0f, 1.0f);
    } else if (metric == 1) {
      p = make_float2(-nr, 1.0f);
    } else {
      p = make_float2(-nr, ss);
    }
    rowp16[r] = p;
  }
}

hipError_t launch_make_scan16(const void* X, int x_half, uint64_t row0, uint64_t n, uint32_t dims, uint32_t ld,
                              uint32_t ld16, int metric, __half* X16, float2* rowp16, unsigned long long* n_unsafe,
                              hipStream_t st) {
  if (n == 0) return hipSuccess;
  const uint64_t max_rows = kMaxWorkItems / 64;  // one wave per row; a dispatch holds < 2^32 work-items
  for (uint64_t r0 = 0; r0 < n; r0 += max_rows) {
    const uint64_t m = n - r0 < max_rows ? n - r0 : max_rows;
    const dim3 grid((uint32_t)((m + 3) / 4));
    if (x_half)
      hipLaunchKernelGGL(make_scan16_kernel<__half>, grid, dim3(256), 0, st, (const __half*)X, row0 + r0, m, dims, ld,
                         ld16, metric, X16, rowp16, n_unsafe);
    else
      hipLaunchKernelGGL(make_scan16_kernel<float>, grid, dim3(256), 0, st, (const float*)X, row0 + r0, m, dims, ld,
                         ld16, metric, X16, rowp16, n_unsafe);
  }
  return hipGetLastError();
}

__global__ __launch_bounds__(64) void prep_queries16_kernel(const float* __restrict__ q_in, uint32_t nq, uint32_t dims,
                                                            uint32_t ld16, int metric, __half* __restrict__ Q16,
                                                            float* __restrict__ qgamma, float2* __restrict__ quv) {
  const uint32_t row = blockIdx.x;
  const int lane = threadIdx.x;
  const uint32_t kts = ld16 >> 5;
  auto put = [&](uint32_t c, float v) {  // stages 0..2 are stored a second time after the last stage
    const __half hv = __float2half_rn(v);
    Q16[scanq16_index(row, c >> 5, c & 31u, ld16)] = hv;
    if (c < 96) Q16[scanq16_index(row, kts + (c >> 5), c & 31u, ld16)] = hv;
  };
  if (row >= nq) {
    for (uint32_t c = lane; c < ld16; c += 64) put(c, 0.0f);
    if (lane == 0) {
      qgamma[row] = 1.0f;
      quv[row] = make_float2(1.0f, 0.0f);
    }
    return;
  }
  const float* in = q_in + (size_t)row * dims;
  float ss = 0.0f;
  for (uint32_t c = lane; c < dims; c += 64) ss += in[c] * in[c];
  ss = wave_sum(ss);
  const bool ok = norm_ok(ss);
  const float beta = ok ? __builtin_sqrtf(ss) : 0.0f;
  const float inv = beta > 0.0f ? 1.0f / beta : 0.0f;
  for (uint32_t c = lane; c < ld16; c += 64) put(c, c < dims ? in[c] * inv : 0.0f);
  if (lane == 0) {
    float g = 1.0f, u = 1.0f, v = 0.0f;
    if (beta > 0.0f) {
      if (metric == 1) {
        g = 1.0f / beta;
        u = beta;
      } else if (metric == 0) {
        g = 0.5f / beta;
        u = 2.0f * beta;
        v = ss;
      }
    }
    if (!ok) u = __builtin_nanf("");
    qgamma[row] = g;
    quv[row] = make_float2(u, v);
  }
}

hipError_t launch_prep_queries16(const float* q_in, uint32_t nq, uint32_t dims, uint32_t ld16, uint32_t q_rows,
                                 int metric, __half* Q16, float* qgamma, float2* quv, hipStream_t st) {
  hipLaunchKernelGGL(prep_queries16_kernel, dim3(q_rows), dim3(64), 0, st, q_in, nq, dims, ld16, metric, Q16, qgamma,
                     quv);
  return hipGetLastError();
}

// ---- int8-MFMA filter scan: scan copy, tile parameters, query preparation (k_flati8.hip) -------------------
// One wave per row.  x^ = x * (1/|x|) in fp32 (the filter's own parallel-sum norm), s = max|x^| / 127,
// xi = clamp(rint(x^ / s), -127, 127), e >= |x^ - s xi| (fp32, rounded up by a relative and an absolute margin).
// Row parameters (A, B, C, D) of the lower bound  S_lower = B*gamma_q + D + C*e_q + A*(s_q*I)  (k_flati8.hip):
//     A = a_r s      B = b_r (1 - 1e-6)      C = a_r (1.0001 + e)      D = a_r (1.0001 e + slack)
// with (a_r, b_r) = cosine (-1, 1), IP (-n_r, 1), L2^2 (-n_r, n_r^2 rounded down), and
//     slack = 4e-6 (fp32 evaluation of S_lower and of x * inv) + 1.5e-7 * d (2.5 d u: the gap between this
//     kernel's norms and the canonical sequential-sum norms the re-rank's cosine distances use).
namespace {
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float i8_slack(uint32_t dims) { return 4e-6f + 1.5e-7f * (float)dims; }
__device__ __forceinline__ float i8_err_up(float e2) { return __builtin_sqrtf(e2) * (1.0f + 1e-4f) + 3e-7f; }
}  // namespace

// tgtA / posn (optional, both or neither; indexed by row - base): the row's step is RAISED so that |A_r| equals tgtA
// exactly (the maximum of its 32-row lane group: "rows of a tile ordered by quantisation step" below) and the row is
// stored at position posn of its tile instead of at its own index.
template <typename XT>
__global__ __launch_bounds__(256) void make_scan8_kernel(const XT* __restrict__ X, uint64_t row0, uint64_t n,
                                                         uint32_t dims, uint32_t ld, uint32_t ld8, int metric,
                                                         int8_t* __restrict__ X8, float4* __restrict__ rowp8,
                                                         unsigned long long* __restrict__ n_unsafe,
                                                         const float* __restrict__ tgtA, const uint8_t* __restrict__ posn,
                                                         uint64_t base) {
  const int lane = threadIdx.x & 63;
  const uint64_t i = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  const uint64_t r = row0 + i;
  const XT* x = X + r * ld;
  float ss = 0.0f;
  for (uint32_t c = lane; c < dims; c += 64) ss += ld_row(x, c) * ld_row(x, c);
  ss = wave_sum(ss);
  const bool ok = norm_ok(ss);
  const float nr = ok ? __builtin_sqrtf(ss) : 0.0f;
  const float inv = nr > 0.0f ? 1.0f / nr : 0.0f;
  float amax = 0.0f;
  for (uint32_t c = lane; c < dims; c += 64) amax = fmaxf(amax, fabsf(ld_row(x, c) * inv));
  amax = wave_max(amax);
  const float a_abs = metric == 2 ? 1.0f : nr;          // |a_r|: cosine 1, inner product and L2^2 the row's norm
  float s = amax / 127.0f;
  if (tgtA) {
    // the step that makes |a_r| s equal to the group's |A| (never below the row's own: tgt is the group maximum of
    // |a_r| amax / 127 computed with these very operations; a rounding of the division downwards is caught here)
    const float want = a_abs > 0.0f ? tgtA[r - base] / a_abs : s;
    s = fmaxf(s, want);
  }
  const float rs = s > 0.0f ? 1.0f / s : 0.0f;
  const uint64_t pr = posn ? ((r & ~(uint64_t)255) | (uint64_t)posn[r - base]) : r;   // where the row is stored
  float e2 = 0.0f;
  for (uint32_t c = lane; c < ld8; c += 64) {
    const float v = c < dims ? ld_row(x, c) * inv : 0.0f;
    float qf = rintf(v * rs);
    qf = fminf(fmaxf(qf, -127.0f), 127.0f);
    X8[scan8_index(pr, c, ld8)] = (int8_t)(int)qf;
    const float res = v - s * qf;
    e2 += res * res;
  }
  e2 = wave_sum(e2);
  if (lane == 0) {
    float4 p;
    if (!ok) {
      p = make_float4(0.0f, __builtin_inff(), 0.0f, 0.0f);
      atomicAdd(n_unsafe, 1ull);
    } else {
      const float e = i8_err_up(e2);
      float a_r = -1.0f, b_r = 1.0f;
      if (metric == 1) {
        a_r = -nr;
      } else if (metric == 0) {
        a_r = -nr;
        b_r = ss * (1.0f - 1e-6f - 7e-8f * ((float)(dims >> 6) + 8.0f));  // the parallel sum, rounded down
      }
      p = make_float4(a_r * s, b_r * (1.0f - 1e-6f), a_r * (1.0001f + e), a_r * (1.0001f * e + i8_slack(dims)));
    }
    rowp8[pr] = p;
  }
}

// natural |A_r| = |a_r| max|x^| / 127 of rows [row0, row0 + n) (what make_scan8_kernel would give the row on its own;
// 0 for a row the filter cannot bound): the sort key of the tile ordering
template <typename XT>
__global__ __launch_bounds__(256) void scan8_step_kernel(const XT* __restrict__ X, uint64_t row0, uint64_t n, uint32_t dims,
                                                         uint32_t ld, int metric, float* __restrict__ natA,
                                                         float* __restrict__ natN) {
  const int lane = threadIdx.x & 63;
  const uint64_t i = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  const XT* x = X + (row0 + i) * ld;
  float ss = 0.0f;
  for (uint32_t c = lane; c < dims; c += 64) ss += ld_row(x, c) * ld_row(x, c);
  ss = wave_sum(ss);
  const bool ok = norm_ok(ss);
  const float nr = ok ? __builtin_sqrtf(ss) : 0.0f;
  const float inv = nr > 0.0f ? 1.0f / nr : 0.0f;
  float amax = 0.0f;
  for (uint32_t c = lane; c < dims; c += 64) amax = fmaxf(amax, fabsf(ld_row(x, c) * inv));
  amax = wave_max(amax);
  if (lane == 0) {
    natA[i] = ok ? (metric == 2 ? 1.0f : nr) * (amax / 127.0f) : 0.0f;
    natN[i] = metric == 0 && ok ? nr : 0.0f;   // the second ordering key: only L2^2 carries the norm in B_r (= |x_r|^2)
  }
}

// (max|A|, max|C|, max|D|, min B) over the 256 rows of each tile: one wave per tile, 4 rows per lane
__global__ __launch_bounds__(64) void tile_params8_kernel(const float4* __restrict__ rowp8, uint64_t tile0,
                                                          float4* __restrict__ tilep8) {
  const int lane = threadIdx.x;
  const uint64_t tile = tile0 + blockIdx.x;
  float am = 0.0f, cm = 0.0f, dm = 0.0f, bm = __builtin_inff();
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float4 p = rowp8[tile * 256 + (uint64_t)j * 64 + lane];
    am = fmaxf(am, fabsf(p.x));
    cm = fmaxf(cm, fabsf(p.z));
    dm = fmaxf(dm, fabsf(p.w));
    bm = fminf(bm, p.y);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    am = fmaxf(am, __shfl_xor(am, o, 64));
    cm = fmaxf(cm, __shfl_xor(cm, o, 64));
    dm = fmaxf(dm, __shfl_xor(dm, o, 64));
    bm = fminf(bm, __shfl_xor(bm, o, 64));
  }
  if (lane == 0) tilep8[tile] = make_float4(am, cm, dm, bm);
}

// ---- rows of a tile ordered by quantisation step (round 4) ---------------------------------------------------
// The scan's epilogue asks, per query and wave tile, whether ANY accumulator can belong to a candidate:
// I * |A_r| >= K_q.  With a per-row |A_r| that is a convert and a multiply per accumulator (13 % of a tile at d = 768,
// more than the matrix work at d = 128).  If the 32 rows a lane holds for one query share (nearly) one step the test is
// an integer maximum and ONE product: max(I) * max|A| >= K_q.  |A_r| varies +-10 % between rows, so the rows of a FULL
// tile are stored ordered by |A_r|: the lane's 32 rows of wave row wr and lane quarter q' = l >> 4 — positions
// 128 wr + 16 rb + 4 q' + r, rb = 0..7, r = 0..3 — hold ranks 32 g .. 32 g + 31, g = 4 wr + q', whose steps differ by a
// few per cent.  perm8[position] = the row's index inside its tile (the kernel maps a hit's position back to the row
// id when it flushes its staging buffer); tileg8[tile][g] = max |A| of group g (8 of the 16 floats of a tile's entry are
// used).  Tiles that are not full when they are written (the tail of a batch, rows appended one by one) keep the
// identity order — their group maxima are the actual maxima, looser but sound — and a tile may only be re-ordered while
// no scan can read it (rows beyond the published row count, or a writer that holds the space exclusively).
namespace {
__device__ __forceinline__ uint32_t i8_group_of_pos(uint32_t p) { return ((p >> 7) << 2) | ((p >> 2) & 3u); }
__device__ __forceinline__ uint32_t i8_pos_of_rank(uint32_t rank) {
  const uint32_t g = rank >> 5, m = rank & 31u;
  return (g >> 2) * 128u + (m >> 2) * 16u + (g & 3u) * 4u + (m & 3u);
}
}  // namespace

// one workgroup per FULL tile (tile index tile0 + blockIdx.x): ranks of the 256 natural |A_r| -> the rows' positions
// (posn, perm8), the eight group maxima (tileg8) and every row's target |A| = the maximum of its group
//
// Round 6 — rows whose NORMS vary (L2^2 on raw rows: B_r = |x_r|^2, the reference's default metric, index.cc:13): the alarm
// level of a lane group is K = (min B gamma_q - ...) / s_q against max |A| of the group, and with ONE min B per tile a tile of
// raw N(0,1) 128-dim rows (norms +-6 %, B +-12 %) alarms in 95 % of its (group, query) pairs against 12 % for normalised rows
// (scripts/studies/int8_alarm_l2_groups.py).  So (i) every lane group gets its own min B (tileg8[tile][8 + g] = min B of the
// group - min B of the tile, k_flati8.hip adds it to the level), and (ii) a tile whose norms spread by more than 0.1 % is
// ordered in FOUR NORM BANDS of 64 rows, each band by step: groups 2 b, 2 b + 1 hold band b's lower and upper half — B
// within a group tight, |A| within a group half as spread: 36 % alarms in the study (norm only: 41 %, step only + group
// B: 64 %).  Tiles of (nearly) equal norms — cosine, inner product, normalised rows — keep the pure step order.
__global__ __launch_bounds__(256) void rank_tiles8_kernel(const float* __restrict__ natA, const float* __restrict__ natN,
                                                          uint64_t tile0, uint8_t* __restrict__ perm8,
                                                          float* __restrict__ tileg8, float* __restrict__ tgtA,
                                                          uint8_t* __restrict__ posn) {
  __shared__ float a_l[256];
  __shared__ float n_l[256];
  __shared__ uint32_t g_l[8];
  __shared__ uint32_t nmm[2];
  const uint32_t tid = threadIdx.x;
  const uint64_t tile = tile0 + blockIdx.x;
  const size_t i = (size_t)blockIdx.x * 256 + tid;   // index into natA / natN / tgtA / posn (rows from tile0 * 256)
  const float a = natA[i];
  const float nrm = natN[i];
  a_l[tid] = a;
  n_l[tid] = nrm;
  if (tid < 8) g_l[tid] = 0u;
  if (tid == 0) {
    nmm[0] = 0x7F800000u;
    nmm[1] = 0u;
  }
  __syncthreads();
  atomicMin(&nmm[0], __float_as_uint(nrm));   // (non-negative floats order like their bit patterns)
  atomicMax(&nmm[1], __float_as_uint(nrm));
  __syncthreads();
  const float nmin = __uint_as_float(nmm[0]), nmax = __uint_as_float(nmm[1]);
  const bool banded = nmax > nmin * 1.001f;
  uint32_t rank = 0;
  if (!banded) {
    for (uint32_t j = 0; j < 256; ++j) {
      const float b = a_l[j];
      rank += (b < a || (b == a && j < tid)) ? 1u : 0u;
    }
  } else {
    uint32_t rn = 0;    // rank by norm -> band
    for (uint32_t j = 0; j < 256; ++j) {
      const float b = n_l[j];
      rn += (b < nrm || (b == nrm && j < tid)) ? 1u : 0u;
    }
    const uint32_t band = rn >> 6;
    __syncthreads();
    n_l[tid] = __uint_as_float(band);   // (n_l now holds every row's band)
    __syncthreads();
    uint32_t rs = 0;    // rank by step inside the band
    for (uint32_t j = 0; j < 256; ++j) {
      const float b = a_l[j];
      const bool same = __float_as_uint(n_l[j]) == band;
      rs += (same && (b < a || (b == a && j < tid))) ? 1u : 0u;
    }
    rank = band * 64u + rs;
  }
  const uint32_t pos = i8_pos_of_rank(rank);
  atomicMax(&g_l[rank >> 5], __float_as_uint(a));   // the group's |A|: the maximum of its rows' natural steps
  __syncthreads();
  const float gmax = __uint_as_float(g_l[rank >> 5]);
  posn[i] = (uint8_t)pos;
  tgtA[i] = gmax;
  perm8[tile * 256 + pos] = (uint8_t)tid;
  if (tid < 8) tileg8[tile * 16 + tid] = __uint_as_float(g_l[tid]);
  else if (tid < 16) tileg8[tile * 16 + tid] = 0.0f;   // (the groups' B margins: ident_tiles8_kernel, from the rows as stored)
}

// tiles kept in row order: perm = identity, group maxima from the rows where they are
// (tiles == nullptr: tiles tile0 + blockIdx.x, perm8 left alone — the ordered tiles, whose group maxima are taken from
// the row parameters as stored too: the alarm's |A| must bound the rows' ACTUAL |A_r| to the last bit)
__global__ __launch_bounds__(256) void ident_tiles8_kernel(const float4* __restrict__ rowp8, uint8_t* __restrict__ perm8,
                                                           float* __restrict__ tileg8, const uint64_t* __restrict__ tiles,
                                                           uint64_t tile0, unsigned long long* __restrict__ n_margin) {
  __shared__ uint32_t gm[8], bm[8];
  const uint32_t tid = threadIdx.x;
  const uint64_t tile = tiles ? tiles[blockIdx.x] : tile0 + blockIdx.x;
  if (tid < 8) {
    gm[tid] = 0u;
    bm[tid] = 0x7F800000u;   // +inf
  }
  __syncthreads();
  const float4 p = rowp8[tile * 256 + tid];
  const uint32_t g = i8_group_of_pos(tid);
  atomicMax(&gm[g], __float_as_uint(fabsf(p.x)));  // (non-negative floats order like their bit patterns)
  // B_r >= 0 for every metric (1 - 1e-6, or |x_r|^2 rounded down; +inf: a padding row or one the filter cannot bound);
  // anything else (never produced) counts as 0: the margin below then is 0, the tile-level bound
  atomicMin(&bm[g], p.y >= 0.0f ? __float_as_uint(p.y) : 0u);
  if (tiles) perm8[tile * 256 + tid] = (uint8_t)tid;
  __syncthreads();
  if (tid < 8) tileg8[tile * 16 + tid] = __uint_as_float(gm[tid]);
  else if (tid < 16) {
    // tileg8[tile][8 + g] = (min B of group g) - (min B of the tile = tilep8[tile].w, tile_params8_kernel), ROUNDED DOWN:
    // the scan's alarm level of the group may assume B_r >= min B of the tile + this margin for every row of the group
    uint32_t tmin = bm[0];
    for (int j = 1; j < 8; ++j) tmin = bm[j] < tmin ? bm[j] : tmin;
    const float bt = __uint_as_float(tmin), bg = __uint_as_float(bm[tid - 8]);
    float margin = 0.0f;
    if (bt < __builtin_inff()) margin = bg < __builtin_inff() ? (bg - bt) * (1.0f - 1e-6f) : __builtin_inff();
    tileg8[tile * 16 + tid] = margin > 0.0f ? margin : 0.0f;
    // (n_margin[1]: tiles where a margin is worth having — the scan only spends instructions on them in spaces that have any)
    if (margin > 1e-3f * bt && margin < __builtin_inff()) atomicAdd(n_margin + 1, 1ull);
  }
}

namespace {
template <typename XT>
void make_rows8(const XT* X, uint64_t row0, uint64_t n, uint32_t dims, uint32_t ld, uint32_t ld8, int metric, int8_t* X8,
                float4* rowp8, unsigned long long* n_unsafe, const float* tgtA, const uint8_t* posn, uint64_t base,
                hipStream_t st) {
  const uint64_t max_rows = kMaxWorkItems / 64;  // one wave per row; a dispatch holds < 2^32 work-items
  for (uint64_t r0 = 0; r0 < n; r0 += max_rows) {
    const uint64_t m = n - r0 < max_rows ? n - r0 : max_rows;
    hipLaunchKernelGGL(make_scan8_kernel<XT>, dim3((uint32_t)((m + 3) / 4)), dim3(256), 0, st, X, row0 + r0, m, dims, ld,
                       ld8, metric, X8, rowp8, n_unsafe, tgtA, posn, base);
  }
}
template <typename XT>
void step_rows8(const XT* X, uint64_t row0, uint64_t n, uint32_t dims, uint32_t ld, int metric, float* natA, float* natN,
                hipStream_t st) {
  const uint64_t max_rows = kMaxWorkItems / 64;
  for (uint64_t r0 = 0; r0 < n; r0 += max_rows) {
    const uint64_t m = n - r0 < max_rows ? n - r0 : max_rows;
    hipLaunchKernelGGL(scan8_step_kernel<XT>, dim3((uint32_t)((m + 3) / 4)), dim3(256), 0, st, X, row0 + r0, m, dims, ld,
                       metric, natA + r0, natN + r0);
  }
}
}  // namespace

size_t make_scan8_scratch_bytes(uint64_t row0, uint64_t n, uint64_t sort_lo, uint64_t sort_hi) {
  const uint64_t t0 = row0 >> 8, t1 = (row0 + n + 255) >> 8;
  const uint64_t s0 = (sort_lo + 255) >> 8, s1 = sort_hi >> 8;
  const uint64_t a0 = s0 > t0 ? s0 : t0, a1 = s1 < t1 ? s1 : t1;
  const uint64_t rows = a1 > a0 ? (a1 - a0) * 256 : 0;
  return (size_t)rows * 13 + (size_t)(t1 - t0 + 1) * 8 + 64;   // natA f32 | tgtA f32 | natN f32 | posn u8 | tile ids
}

hipError_t launch_make_scan8(const void* X, int x_half, uint64_t row0, uint64_t n, uint32_t dims, uint32_t ld,
                             uint32_t ld8, int metric, int8_t* X8, float4* rowp8, float4* tilep8, uint8_t* perm8,
                             float* tileg8, uint64_t sort_lo, uint64_t sort_hi, void* scratch,
                             unsigned long long* n_unsafe, hipStream_t st) {
  if (n == 0) return hipSuccess;
  const uint64_t t0 = row0 >> 8, t1 = (row0 + n + 255) >> 8;
  // tiles wholly inside [sort_lo, sort_hi) AND inside the rows being written are ordered by step; the others keep the
  // row order
  const uint64_t s0 = (sort_lo + 255) >> 8, s1 = sort_hi >> 8;
  uint64_t a0 = s0 > t0 ? s0 : t0, a1 = s1 < t1 ? s1 : t1;
  if (a0 * 256 < row0) a0 = (row0 + 255) >> 8;
  if (a1 * 256 > row0 + n) a1 = (row0 + n) >> 8;
  const uint64_t n_sorted = a1 > a0 ? a1 - a0 : 0;
  auto make = [&](uint64_t r0, uint64_t m, const float* tgt, const uint8_t* pos, uint64_t base) {
    if (m == 0) return;
    if (x_half) make_rows8((const __half*)X, r0, m, dims, ld, ld8, metric, X8, rowp8, n_unsafe, tgt, pos, base, st);
    else make_rows8((const float*)X, r0, m, dims, ld, ld8, metric, X8, rowp8, n_unsafe, tgt, pos, base, st);
  };
  uint64_t* tile_list;
  if (n_sorted) {
    const uint64_t rows = n_sorted * 256, base = a0 * 256;
    float* natA = (float*)scratch;
    float* tgtA = natA + rows;
    float* natN = tgtA + rows;
    uint8_t* posn = (uint8_t*)(natN + rows);
    tile_list = (uint64_t*)(((uintptr_t)(posn + rows) + 15) & ~(uintptr_t)15);
    if (x_half) step_rows8((const __half*)X, base, rows, dims, ld, metric, natA, natN, st);
    else step_rows8((const float*)X, base, rows, dims, ld, metric, natA, natN, st);
    for (uint64_t c0 = 0; c0 < n_sorted; c0 += 1u << 30) {
      const uint64_t m = n_sorted - c0 < (1u << 30) ? n_sorted - c0 : (1u << 30);
      hipLaunchKernelGGL(rank_tiles8_kernel, dim3((uint32_t)m), dim3(256), 0, st, natA + c0 * 256, natN + c0 * 256, a0 + c0,
                         perm8, tileg8, tgtA + c0 * 256, posn + c0 * 256);
    }
    make(row0, base - row0, nullptr, nullptr, 0);                       // rows before the ordered tiles
    make(base, rows, tgtA, posn, base);                                 // the ordered tiles
    make(a1 * 256, row0 + n - a1 * 256, nullptr, nullptr, 0);           // rows after them
  } else {
    tile_list = (uint64_t*)scratch;
    make(row0, n, nullptr, nullptr, 0);
  }
  hipLaunchKernelGGL(tile_params8_kernel, dim3((uint32_t)(t1 - t0)), dim3(64), 0, st, rowp8, t0, tilep8);
  const uint64_t n_ident = (t1 - t0) - n_sorted;
  if (n_ident) {
    // ids of the tiles of [t0, t1) outside [a0, a1)
    if (hipError_t e = launch_tile_ids(tile_list, t0, t1, n_sorted ? a0 : t1, n_sorted ? a1 : t1, st); e != hipSuccess) return e;
    hipLaunchKernelGGL(ident_tiles8_kernel, dim3((uint32_t)n_ident), dim3(256), 0, st, rowp8, perm8, tileg8,
                       tile_list + n_sorted, (uint64_t)0, n_unsafe);
  }
  for (uint64_t c0 = 0; c0 < n_sorted; c0 += 1u << 30) {
    const uint64_t m = n_sorted - c0 < (1u << 30) ? n_sorted - c0 : (1u << 30);
    hipLaunchKernelGGL(ident_tiles8_kernel, dim3((uint32_t)m), dim3(256), 0, st, rowp8, perm8, tileg8,
                       (const uint64_t*)nullptr, a0 + c0, n_unsafe);
  }
  return hipGetLastError();
}

// tile_list[0 .. a1-a0) = a0 .. a1-1 (the sorted tiles), then the tiles of [t0, t1) outside [a0, a1)
__global__ __launch_bounds__(256) void tile_ids_kernel(uint64_t* __restrict__ out, uint64_t t0, uint64_t t1, uint64_t a0,
                                                       uint64_t a1) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  const uint64_t ns = a1 - a0, total = t1 - t0;
  if (i >= total) return;
  if (i < ns) out[i] = a0 + i;
  else {
    const uint64_t j = i - ns;                  // j-th tile of [t0, a0) ++ [a1, t1)
    out[i] = j < a0 - t0 ? t0 + j : a1 + (j - (a0 - t0));
  }
}
hipError_t launch_tile_ids(uint64_t* out, uint64_t t0, uint64_t t1, uint64_t a0, uint64_t a1, hipStream_t st) {
  if (t1 <= t0) return hipSuccess;
  hipLaunchKernelGGL(tile_ids_kernel, dim3((uint32_t)((t1 - t0 + 255) / 256)), dim3(256), 0, st, out, t0, t1, a0, a1);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void perm8_pad_kernel(uint8_t* __restrict__ perm8, uint64_t row0, uint64_t n) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) perm8[row0 + i] = (uint8_t)((row0 + i) & 255u);
}
hipError_t launch_perm8_pad(uint8_t* perm8, uint64_t row0, uint64_t n, hipStream_t st) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(perm8_pad_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, perm8, row0, n);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void rowp8_pad_kernel(float4* __restrict__ rowp8, uint64_t row0, uint64_t n) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) rowp8[row0 + i] = make_float4(0.0f, __builtin_inff(), 0.0f, 0.0f);
}
__global__ __launch_bounds__(256) void tilep8_pad_kernel(float4* __restrict__ tilep8, uint64_t t0, uint64_t n) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) tilep8[t0 + i] = make_float4(0.0f, 0.0f, 0.0f, __builtin_inff());
}
hipError_t launch_rowp8_pad(float4* rowp8, uint64_t row0, uint64_t n, hipStream_t st) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(rowp8_pad_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, rowp8, row0, n);
  return hipGetLastError();
}
hipError_t launch_tilep8_pad(float4* tilep8, uint64_t t0, uint64_t n, hipStream_t st) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(tilep8_pad_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, tilep8, t0, n);
  return hipGetLastError();
}

// Everything a batch of the int8 engine needs before its first scan launch, ONE launch (it was three: the fp32 query
// rows of the re-rank, the int8 tiles + parameters, and a memset of the scan's control words).
// ctl = [q_rows] pool counts | [q_rows] overflow flags | [kSyncWordsI8] counters.
//
// Round 6: one workgroup of TWO waves per query row.  The row is read once, coalesced, into LDS; wave 0 makes the prepared
// fp32 row of the re-rank (cosine: the canonical norm is ONE sequential sum in hnswlib's order — lane 0 walks the row in LDS,
// the same multiplications and additions as prep_query_row / seq_sumsq, bit for bit), wave 1 meanwhile the int8 tile and the
// bound's parameters: every lane owns whole 16-byte chunks of the stage-blocked layout (16 columns: four LDS reads, ONE
// 16-byte store per copy instead of sixteen byte stores).  Before: one wave per row did both in turn, read the row four
// times from global memory and wrote the int8 tile a byte at a time — 27 us at 1024 x 768, 3 % of a 1 M-row batch.
__global__ __launch_bounds__(128) void prep_queries_i8_kernel(const float* __restrict__ q_in, uint32_t nq, uint32_t dims,
                                                              uint32_t ld, uint32_t ld8, uint32_t q_rows, int metric,
                                                              float* __restrict__ q_out, int8_t* __restrict__ Q8,
                                                              float4* __restrict__ qparams, float2* __restrict__ quv,
                                                              float* __restrict__ thr, uint32_t* __restrict__ ctl) {
  __shared__ __attribute__((aligned(16))) float xs[2048 + 64];   // the raw row, zero beyond dims (dims <= 2048: ehx_api.cpp)
  const uint32_t row = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const bool live = row < nq;
  const uint32_t cmax = (ld8 > ld ? ld8 : ld);   // (<= 2048 + padding of the fp32 stride)
  {
    const float* in = q_in + (size_t)row * dims;
    for (uint32_t c = (uint32_t)tid; c < cmax && c < 2048u + 64u; c += 128) xs[c] = (live && c < dims) ? in[c] : 0.0f;
  }
  __syncthreads();
  if (w == 0) {
    // ---- the prepared fp32 row ----
    float* out = q_out + (size_t)row * ld;
    float inv = 1.0f;
    if (live && metric == 2) {
      float v = 0.0f;
      if (lane == 0) {
        float sum = 0.0f;
        uint32_t i = 0;
        // (32 elements per trip, all eight LDS reads issued before the first addition: the chain of 768 dependent additions
        // is the floor of this kernel, an LDS round trip per four of them on top was three times that)
        for (; i + 32 <= dims; i += 32) {
          float4 b[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) b[j] = *(const float4*)(xs + i + 4 * j);
#pragma unroll
          for (int j = 0; j < 8; ++j)
            sum = ex_add(ex_add(ex_add(ex_add(sum, ex_mul(b[j].x, b[j].x)), ex_mul(b[j].y, b[j].y)), ex_mul(b[j].z, b[j].z)),
                         ex_mul(b[j].w, b[j].w));
        }
        for (; i + 4 <= dims; i += 4) {
          const float4 x4 = *(const float4*)(xs + i);
          sum = ex_add(ex_add(ex_add(ex_add(sum, ex_mul(x4.x, x4.x)), ex_mul(x4.y, x4.y)), ex_mul(x4.z, x4.z)), ex_mul(x4.w, x4.w));
        }
        for (; i < dims; ++i) sum = ex_add(sum, ex_mul(xs[i], xs[i]));
        v = inv_norm_of(sum);
      }
      inv = __shfl(v, 0, 64);
    }
    for (uint32_t i = (uint32_t)lane; i < ld; i += 64) {
      float v = i < 2048u + 64u ? xs[i] : 0.0f;
      if (live && metric == 2) v = ex_mul(v, inv);
      out[i] = v;
    }
    if (lane == 0) {
      ctl[row] = 0;
      ctl[q_rows + row] = 0;
    }
    if (row == 0)
      for (uint32_t i = lane; i < kSyncWordsI8; i += 64) ctl[2 * (size_t)q_rows + i] = 0;
    return;
  }
  // ---- wave 1: the int8 tile (scanq8_index: [stage][row of the query tile][64 bytes, 16-byte chunks swizzled by the row];
  // the three blocks after the last stage repeat the tile's first stages — block kts + j holds stage j mod kts — so that the
  // scan's three-stage look-ahead reads linearly across a tile boundary) and the parameters ----
  const uint32_t kts = ld8 >> 6, n_chunks = kts * 4u;   // (<= 128: two chunks per lane at most)
  auto put16 = [&](uint32_t chunk, const int4 v) {
    const uint32_t stg = chunk >> 2, cl = chunk & 3u;
    // (scanq8_index(row, stage, 16 cl, ld8): the chunk's 16 bytes are contiguous and 16-byte aligned)
    *(int4*)(Q8 + scanq8_index(row, stg, cl * 16u, ld8)) = v;
    for (uint32_t j = stg; j < 3u; j += kts) *(int4*)(Q8 + scanq8_index(row, kts + j, cl * 16u, ld8)) = v;
  };
  if (!live) {
    for (uint32_t ch = (uint32_t)lane; ch < n_chunks; ch += 64) put16(ch, make_int4(0, 0, 0, 0));
    if (lane == 0) {
      qparams[row] = make_float4(0.0f, 0.0f, 1.0f, __builtin_inff());
      quv[row] = make_float2(1.0f, 0.0f);
      thr[row] = -__builtin_inff();  // a padding query never collects anything
    }
    return;
  }
  float x[2][16];
  float ss = 0.0f;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const uint32_t ch = (uint32_t)lane + 64u * (uint32_t)t;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      if (ch < n_chunks) v = *(const float4*)(xs + ch * 16u + 4u * (uint32_t)j);
      x[t][4 * j + 0] = v.x;
      x[t][4 * j + 1] = v.y;
      x[t][4 * j + 2] = v.z;
      x[t][4 * j + 3] = v.w;
      ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
  }
  ss = wave_sum(ss);
  const bool ok = norm_ok(ss);
  const float beta = ok ? __builtin_sqrtf(ss) : 0.0f;
  const float inv = beta > 0.0f ? 1.0f / beta : 0.0f;
  float amax = 0.0f;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int j = 0; j < 16; ++j) amax = fmaxf(amax, fabsf(x[t][j] * inv));
  amax = wave_max(amax);
  const float s = amax / 127.0f;
  const float rs = amax > 0.0f ? 127.0f / amax : 0.0f;
  float e2 = 0.0f;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const uint32_t ch = (uint32_t)lane + 64u * (uint32_t)t;
    uint32_t pk[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const float v = x[t][j] * inv;   // (columns beyond dims: 0 -> code 0, residual 0)
      float qf = rintf(v * rs);
      qf = fminf(fmaxf(qf, -127.0f), 127.0f);
      pk[j >> 2] |= ((uint32_t)(int)qf & 0xFFu) << (8 * (j & 3));
      const float res = v - s * qf;
      e2 += res * res;
    }
    if (ch < n_chunks) put16(ch, make_int4((int)pk[0], (int)pk[1], (int)pk[2], (int)pk[3]));
  }
  e2 = wave_sum(e2);
  if (lane == 0) {
    float g = 1.0f, u = 1.0f, v = 0.0f;
    if (beta > 0.0f) {
      if (metric == 1) {
        g = 1.0f / beta;
        u = beta;
      } else if (metric == 0) {
        g = 0.5f / beta;
        u = 2.0f * beta;
        v = ss * (1.0f - 1e-6f - 7e-8f * ((float)(dims >> 6) + 8.0f));  // |q|^2 rounded down
      }
    }
    if (!ok) u = __builtin_nanf("");
    qparams[row] = make_float4(s, i8_err_up(e2), g, __builtin_inff());  // .w: smallest threshold used so far
    quv[row] = make_float2(u, v);
    thr[row] = __builtin_inff();
  }
}

hipError_t launch_prep_queries_i8(const float* q_in, uint32_t nq, uint32_t dims, uint32_t ld, uint32_t ld8,
                                  uint32_t q_rows, int metric, float* q_out, int8_t* Q8, float4* qparams, float2* quv,
                                  float* thr, uint32_t* ctl, hipStream_t st) {
  if (dims > 2048u || ld8 > 2048u || ld > 2048u + 64u) return hipErrorInvalidValue;   // (the int8 engine's own limit: ehx_api.cpp)
  hipLaunchKernelGGL(prep_queries_i8_kernel, dim3(q_rows), dim3(128), 0, st, q_in, nq, dims, ld, ld8, q_rows, metric,
                     q_out, Q8, qparams, quv, thr, ctl);
  return hipGetLastError();
}

// ---- fp16 storage ----------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void store_rows_f16_kernel(const float* __restrict__ src, uint32_t src_ld,
                                                             const uint64_t* __restrict__ ids, uint64_t row0,
                                                             uint64_t n, uint32_t dims, uint32_t ld,
                                                             __half* __restrict__ X) {
  const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= n * ld) return;
  const uint64_t i = gid / ld;
  const uint32_t c = (uint32_t)(gid - i * ld);
  const uint64_t r = ids ? ids[i] : row0 + i;
  X[r * ld + c] = c < dims ? __float2half_rn(src[i * src_ld + c]) : __float2half_rn(0.0f);
}

hipError_t launch_store_rows_f16(const float* src, uint32_t src_ld, const uint64_t* ids, uint64_t row0, uint64_t n,
                                 uint32_t dims, uint32_t ld, __half* X, hipStream_t st) {
  if (n == 0) return hipSuccess;
  const uint64_t work = n * ld;
  hipLaunchKernelGGL(store_rows_f16_kernel, dim3((uint32_t)((work + 255) / 256)), dim3(256), 0, st, src, src_ld, ids,
                     row0, n, dims, ld, X);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void load_row_f16_kernel(const __half* __restrict__ X, uint64_t row, uint32_t dims,
                                                           uint32_t ld, float* __restrict__ out) {
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < dims) out[c] = __half2float(X[row * ld + c]);
}

hipError_t launch_load_row_f16(const __half* X, uint64_t row, uint32_t dims, uint32_t ld, float* out, hipStream_t st) {
  hipLaunchKernelGGL(load_row_f16_kernel, dim3((dims + 255) / 256), dim3(256), 0, st, X, row, dims, ld, out);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void rowp_pad_kernel(float2* __restrict__ rowp, uint64_t row0, uint64_t n) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) rowp[row0 + i] = make_float2(0.0f, __builtin_inff());
}

hipError_t launch_rowp_pad(float2* rowp, uint64_t row0, uint64_t n, hipStream_t st) {
  if (n == 0) return hipSuccess;
  const uint32_t grid = (uint32_t)((n + 255) / 256);
  hipLaunchKernelGGL(rowp_pad_kernel, dim3(grid), dim3(256), 0, st, rowp, row0, n);
  return hipGetLastError();
}

// ---- EHX-GAUSS-1 --------------------------------------------------------------------------------
// pass 1: one thread per 4 columns (one Philox call), coalesced 16-B stores
__global__ __launch_bounds__(256) void gen_rows_kernel(uint64_t seed, uint64_t row0, uint64_t row_stride, uint64_t n_rows,
                                                       uint32_t dims, uint32_t ld, float* __restrict__ out) {
  const uint32_t cbs = ld / 4;  // ld % 4 == 0
  const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= n_rows * cbs) return;
  const uint64_t r = gid / cbs;
  const uint32_t cb = (uint32_t)(gid - r * cbs);
  float z[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  if (cb * 4 < dims) {
    ehx_datagen::normal4(seed, row0 + r * row_stride, cb, z);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (cb * 4 + j >= dims) z[j] = 0.0f;
  }
  *(float4*)(out + r * ld + cb * 4) = make_float4(z[0], z[1], z[2], z[3]);
}

// pass 2 (normalize): one wave per row; lane 0 walks the row for the canonical sum
__global__ __launch_bounds__(64) void normalize_rows_kernel(uint64_t n_rows, uint32_t dims, uint32_t ld,
                                                            float* __restrict__ x) {
  const uint64_t r = blockIdx.x;
  const int lane = threadIdx.x;
  float* row = x + r * ld;
  float v = 0.0f;
  if (lane == 0) v = inv_norm_of(seq_sumsq(row, dims));
  const float inv = __shfl(v, 0, 64);
  for (uint32_t i = lane; i < dims; i += 64) row[i] = ex_mul(row[i], inv);
}

// faster variant for big fills: 64 rows per wave, 64x64 tiles transposed through LDS so global
// accesses stay coalesced while each lane still sums ITS row sequentially (same order, same bits)
__global__ __launch_bounds__(64) void normalize_rows_tiled_kernel(uint64_t n_rows, uint32_t dims, uint32_t ld,
                                                                  float* __restrict__ x) {
  __shared__ float tile[64][65];
  const int lane = threadIdx.x;
  const uint64_t r0 = (uint64_t)blockIdx.x * 64;
  const uint64_t rows = n_rows - r0 < 64 ? n_rows - r0 : 64;
  float s = 0.0f;
  for (uint32_t c0 = 0; c0 < dims; c0 += 64) {
    for (uint32_t rr = 0; rr < rows; ++rr) {
      const uint32_t c = c0 + lane;
      tile[rr][lane] = c < dims ? x[(r0 + rr) * ld + c] : 0.0f;
    }
    __syncthreads();
    if ((uint64_t)lane < rows) {
      const uint32_t lim = dims - c0 < 64 ? dims - c0 : 64;
      for (uint32_t c = 0; c < lim; ++c) s = ex_add(s, ex_mul(tile[lane][c], tile[lane][c]));
    }
    __syncthreads();
  }
  const float inv = inv_norm_of(s);
  for (uint32_t rr = 0; rr < rows; ++rr) {
    const float rinv = __shfl(inv, (int)rr, 64);
    for (uint32_t c = lane; c < dims; c += 64) {
      float* p = x + (r0 + rr) * ld + c;
      *p = ex_mul(*p, rinv);
    }
  }
}

// ---- EHX-MANIFOLD-1 (include/ehx_datagen.h): rows on an R-dimensional linear subspace + 5 % noise ------------------
// the basis b[j][c] (R x ld floats, padding columns 0), once per (device, R, dims, ld): one thread per 4 columns
__global__ __launch_bounds__(256) void manifold_basis_kernel(uint32_t R, uint32_t dims, uint32_t ld, float* __restrict__ basis) {
  const uint32_t cbs = ld / 4;
  const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= R * cbs) return;
  const uint32_t j = gid / cbs, cb = gid - j * cbs;
  float b[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  if (cb * 4 < dims) {
    ehx_datagen::manifold_basis4(j, cb, R, b);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (cb * 4 + i >= dims) b[i] = 0.0f;
  }
  *(float4*)(basis + (size_t)j * ld + cb * 4) = make_float4(b[0], b[1], b[2], b[3]);
}

// rows: a workgroup takes kManifoldRows rows — their latents once into LDS, then one thread per (row, 4 columns): the
// noise block (one Philox call, the EHX-GAUSS-1 element), R basis values per column from the L2-resident basis, the
// sequential non-fused sum of the spec
constexpr uint32_t kManifoldRows = 8;
__global__ __launch_bounds__(256) void gen_manifold_rows_kernel(uint64_t seed, uint64_t row0, uint64_t row_stride, uint64_t n_rows,
                                                                uint32_t dims, uint32_t ld, uint32_t R,
                                                                const float* __restrict__ basis, float* __restrict__ out) {
  __shared__ float lat[kManifoldRows][EHX_MANIFOLD_MAX_LATENT];
  const uint64_t r0 = (uint64_t)blockIdx.x * kManifoldRows;
  const uint32_t rows = (uint32_t)(n_rows - r0 < kManifoldRows ? n_rows - r0 : kManifoldRows);
  const uint32_t jbs = (R + 3) / 4;
  for (uint32_t t = threadIdx.x; t < rows * jbs; t += blockDim.x) {
    const uint32_t rr = t / jbs, jb = t - rr * jbs;
    float l[4];
    ehx_datagen::manifold_latent4(seed, row0 + (r0 + rr) * row_stride, jb, l);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (jb * 4 + i < EHX_MANIFOLD_MAX_LATENT) lat[rr][jb * 4 + i] = l[i];
  }
  __syncthreads();
  const uint32_t cbs = ld / 4;
  for (uint32_t t = threadIdx.x; t < rows * cbs; t += blockDim.x) {
    const uint32_t rr = t / cbs, cb = t - rr * cbs;
    float x[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (cb * 4 < dims) {
      float e[4];
      ehx_datagen::normal4(seed, row0 + (r0 + rr) * row_stride, cb, e);
      float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
      for (uint32_t j = 0; j < R; ++j) {
        const float4 b = *(const float4*)(basis + (size_t)j * ld + cb * 4);
        const float lj = lat[rr][j];
        acc[0] = ex_add(acc[0], ex_mul(lj, b.x));
        acc[1] = ex_add(acc[1], ex_mul(lj, b.y));
        acc[2] = ex_add(acc[2], ex_mul(lj, b.z));
        acc[3] = ex_add(acc[3], ex_mul(lj, b.w));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) x[i] = cb * 4 + i < dims ? ex_add(acc[i], ex_mul(0.05f, e[i])) : 0.0f;
    }
    *(float4*)(out + (r0 + rr) * ld + cb * 4) = make_float4(x[0], x[1], x[2], x[3]);
  }
}

namespace {
// one basis per (device, R, dims, ld), made on first use and kept (R * ld * 4 bytes: 49 KB at R = 16, d = 768)
struct BasisKey {
  int dev;
  uint32_t R, dims, ld;
  bool operator<(const BasisKey& o) const {
    if (dev != o.dev) return dev < o.dev;
    if (R != o.R) return R < o.R;
    if (dims != o.dims) return dims < o.dims;
    return ld < o.ld;
  }
};
std::mutex g_basis_mu;
std::map<BasisKey, float*> g_basis;
hipError_t manifold_basis(uint32_t R, uint32_t dims, uint32_t ld, hipStream_t st, const float** out) {
  int dev = 0;
  if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
  std::lock_guard<std::mutex> g(g_basis_mu);
  const BasisKey key{dev, R, dims, ld};
  auto it = g_basis.find(key);
  if (it != g_basis.end()) {
    *out = it->second;
    return hipSuccess;
  }
  float* b = nullptr;
  if (hipError_t e = hipMalloc((void**)&b, (size_t)R * ld * sizeof(float)); e != hipSuccess) return e;
  const uint32_t work = R * (ld / 4);
  hipLaunchKernelGGL(manifold_basis_kernel, dim3((work + 255) / 256), dim3(256), 0, st, R, dims, ld, b);
  if (hipError_t e = hipStreamSynchronize(st); e != hipSuccess) {   // (other streams may use it from now on)
    (void)hipFree(b);
    return e;
  }
  g_basis[key] = b;
  *out = b;
  return hipSuccess;
}
}  // namespace

hipError_t launch_gen_rows(uint64_t seed, uint64_t row0, uint64_t n_rows, uint32_t dims, uint32_t ld,
                           int normalize, float* out, hipStream_t st, uint64_t row_stride, uint32_t latent) {
  if (n_rows == 0) return hipSuccess;
  if (latent) {   // EHX-MANIFOLD-1
    const float* basis = nullptr;
    if (hipError_t e = manifold_basis(latent, dims, ld, st, &basis); e != hipSuccess) return e;
    const uint64_t max_rows = (uint64_t)(1u << 30) * kManifoldRows;
    for (uint64_t done = 0; done < n_rows;) {
      const uint64_t rows = n_rows - done < max_rows ? n_rows - done : max_rows;
      hipLaunchKernelGGL(gen_manifold_rows_kernel, dim3((uint32_t)((rows + kManifoldRows - 1) / kManifoldRows)), dim3(256), 0, st,
                         seed, row0 + done * row_stride, row_stride, rows, dims, ld, latent, basis, out + done * ld);
      done += rows;
    }
    if (normalize)
      hipLaunchKernelGGL(normalize_rows_tiled_kernel, dim3((uint32_t)((n_rows + 63) / 64)), dim3(64), 0, st, n_rows, dims, ld,
                         out);
    return hipGetLastError();
  }
  const uint64_t work = n_rows * (ld / 4);
  // grid.x limit: chunk the launch if needed
  const uint64_t max_blocks = 1u << 30;
  uint64_t done_rows = 0;
  while (done_rows < n_rows) {
    uint64_t rows = n_rows - done_rows;
    const uint64_t per_row_blocks_x256 = ld / 4;  // threads per row
    const uint64_t max_rows = (max_blocks * 256) / per_row_blocks_x256;
    if (rows > max_rows) rows = max_rows;
    const uint64_t threads = rows * per_row_blocks_x256;
    hipLaunchKernelGGL(gen_rows_kernel, dim3((uint32_t)((threads + 255) / 256)), dim3(256), 0, st, seed,
                       row0 + done_rows * row_stride, row_stride, rows, dims, ld, out + done_rows * ld);
    done_rows += rows;
  }
  (void)work;
  if (normalize) {
    hipLaunchKernelGGL(normalize_rows_tiled_kernel, dim3((uint32_t)((n_rows + 63) / 64)), dim3(64), 0, st,
                       n_rows, dims, ld, out);
  }
  return hipGetLastError();
}

// The queries a stage could not certify go to the next engine as a dense sub-batch: one launch gathers their raw rows
// (idx[j] = index in the caller's batch), one launch scatters the sub-batch's results back.  (Round 3 issued one
// hipMemcpyAsync per query for the gather and three per query for the scatter: 4756 copies for the 1189 queries a
// 12.5 M x 1536 batch lost while its candidate list was still 256 wide.)
__global__ void gather_queries_kernel(const float* __restrict__ src, const uint32_t* __restrict__ idx, uint32_t dims,
                                      float* __restrict__ dst) {
  const float* s = src + (size_t)idx[blockIdx.x] * dims;
  float* d = dst + (size_t)blockIdx.x * dims;
  for (uint32_t c = threadIdx.x; c < dims; c += blockDim.x) d[c] = s[c];
}

__global__ void scatter_results_kernel(const uint64_t* __restrict__ ids, const float* __restrict__ dist,
                                       const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ idx, uint32_t k,
                                       uint64_t* __restrict__ out_ids, float* __restrict__ out_dist,
                                       uint32_t* __restrict__ out_cnt) {
  const uint32_t j = blockIdx.x, g = idx[j];
  for (uint32_t c = threadIdx.x; c < k; c += blockDim.x) {
    out_ids[(size_t)g * k + c] = ids[(size_t)j * k + c];
    out_dist[(size_t)g * k + c] = dist[(size_t)j * k + c];
  }
  if (threadIdx.x == 0) out_cnt[g] = cnt[j];
}

hipError_t launch_gather_queries(const float* src, const uint32_t* idx, uint32_t m, uint32_t dims, float* dst,
                                 hipStream_t st) {
  if (m == 0) return hipSuccess;
  hipLaunchKernelGGL(gather_queries_kernel, dim3(m), dim3(256), 0, st, src, idx, dims, dst);
  return hipGetLastError();
}

hipError_t launch_scatter_results(const uint64_t* ids, const float* dist, const uint32_t* cnt, const uint32_t* idx,
                                  uint32_t m, uint32_t k, uint64_t* out_ids, float* out_dist, uint32_t* out_cnt,
                                  hipStream_t st) {
  if (m == 0) return hipSuccess;
  hipLaunchKernelGGL(scatter_results_kernel, dim3(m), dim3(64), 0, st, ids, dist, cnt, idx, k, out_ids, out_dist, out_cnt);
  return hipGetLastError();
}

}  // namespace ehx
