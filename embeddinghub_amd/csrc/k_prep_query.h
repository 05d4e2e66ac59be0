// Query preparation shared by prep_queries_kernel (k_misc.hip) and the graph search kernel's one-launch form
// (k_graph.hip): pad to the row stride, cosine queries normalised with the canonical (sequential) norm.
#pragma once
#include "ehx_kernels.h"

namespace ehx {

__device__ __forceinline__ float inv_norm_of(float sumsq) {
  return ex_div(1.0f, ex_add(ex_sqrt(sumsq), 1e-30f));
}

// ((...((0 + s[0]) + s[1]) + ...) + s[n - 1]): ONE sequential sum of n floats in LDS (s 16-byte aligned), exactly in index
// order.  32 elements per trip, the eight LDS reads issued before the first addition: the dependent additions are the floor
// (768 of them ~3 us), an LDS round trip per four of them on top was three times that (round 6: 1024 x 768 queries prepared
// in 13 us instead of 27; one query per call on a graph space 10 us less).
__device__ __forceinline__ float seq_sum_lds(const float* s, uint32_t n) {
  float sum = 0.0f;
  uint32_t i = 0;
  for (; i + 32 <= n; i += 32) {
    float4 b[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) b[j] = *(const float4*)(s + i + 4 * j);
#pragma unroll
    for (int j = 0; j < 8; ++j) sum = ex_add(ex_add(ex_add(ex_add(sum, b[j].x), b[j].y), b[j].z), b[j].w);
  }
  for (; i + 4 <= n; i += 4) {
    const float4 v = *(const float4*)(s + i);
    sum = ex_add(ex_add(ex_add(ex_add(sum, v.x), v.y), v.z), v.w);
  }
  for (; i < n; ++i) sum = ex_add(sum, s[i]);
  return sum;
}

// one wave per output row: lane 0 computes the canonical norm, all lanes scale/copy
__device__ __forceinline__ void prep_query_row(const float* __restrict__ q_in, uint32_t nq, uint32_t dims, uint32_t ld,
                                               int metric, float* __restrict__ q_out, uint32_t row, int lane) {
  float* out = q_out + (size_t)row * ld;
  if (row >= nq) {
    for (uint32_t i = lane; i < ld; i += 64) out[i] = 0.0f;
    return;
  }
  const float* in = q_in + (size_t)row * dims;
  float inv = 1.0f;
  if (metric == 2) {
    // the canonical norm is ONE sequential sum (hnswlib's order): the wave squares 1024 elements at a time into LDS
    // with coalesced loads, lane 0 adds them up in order — the same additions as seq_sumsq, without one lane waiting
    // for 768 global loads eight at a time (same-box A/B at 1 M x 768: 1.161 / 1.160 -> 1.155 / 1.154 ms per batch,
    // profiles/r03_q_prep_queries_lds_norm_ab.jsonl)
    __shared__ __attribute__((aligned(16))) float sq[1024];
    float sum = 0.0f;
    for (uint32_t base = 0; base < dims; base += 1024) {
      const uint32_t m = dims - base < 1024u ? dims - base : 1024u;
      for (uint32_t i = lane; i < m; i += 64) {
        const float v = in[base + i];
        sq[i] = ex_mul(v, v);
      }
      __syncthreads();
      if (lane == 0) {
        uint32_t i = 0;
        for (; i + 32 <= m; i += 32) {   // (seq_sum_lds's trip, continuing the running sum across 1024-element blocks)
          float4 b[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) b[j] = *(const float4*)(sq + i + 4 * j);
#pragma unroll
          for (int j = 0; j < 8; ++j) sum = ex_add(ex_add(ex_add(ex_add(sum, b[j].x), b[j].y), b[j].z), b[j].w);
        }
        for (; i + 4 <= m; i += 4) {
          const float4 v = *(const float4*)(sq + i);
          sum = ex_add(ex_add(ex_add(ex_add(sum, v.x), v.y), v.z), v.w);
        }
        for (; i < m; ++i) sum = ex_add(sum, sq[i]);
      }
      __syncthreads();
    }
    float v = 0.0f;
    if (lane == 0) v = inv_norm_of(sum);
    inv = __shfl(v, 0, 64);
  }
  for (uint32_t i = lane; i < ld; i += 64) {
    float v = i < dims ? in[i] : 0.0f;
    if (metric == 2) v = ex_mul(v, inv);
    out[i] = v;
  }
}

}  // namespace ehx
