// Query preparation shared by prep_queries_kernel (k_misc.hip) and the graph search kernel's one-launch form
// (k_graph.hip): pad to the row stride, cosine queries normalised with the canonical (sequential) norm.
#pragma once
#include "ehx_kernels.h"

namespace ehx {

__device__ __forceinline__ float inv_norm_of(float sumsq) {
  return ex_div(1.0f, ex_add(ex_sqrt(sumsq), 1e-30f));
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
