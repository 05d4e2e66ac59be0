// Flat (exhaustive) kNN: what surrounds the scan kernels (k_flati8.hip, k_flat16.hip, k_flat8.hip) —
//   flat_merge_kernel     per-chunk sorted key lists -> the query's running best 64
//   sample_select_kernel  first thresholds of the fp16 filter's cascade
//   rerank_kernel         canonical (oracle-order) fp32 distances of the candidates, top-k, certificate
//   exhaustive_kernel     the canonical distance of EVERY row (last engine of the chain; paged k > 48)
//   merge_lists_kernel    k-way merge of per-shard results (multi-GPU)
// The scan replaces, for the brute-force configuration, the distance loop the reference runs inside hnswlib
// (searchKnn -> fstdistfunc_, call site embeddinghub/embeddingstore/index.cc:41); the re-rank recomputes the
// surviving distances in exactly the oracle's (hnswlib SSE) summation order, so ids and distances are
// bit-identical to the exhaustive oracle.
#include <cstdlib>

#include "ehx_kernels.h"
#include "k_prep_query.h"

namespace ehx {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

// ascending bitonic sort of one u64 per lane across the 64-lane wave
__device__ __forceinline__ uint64_t wave_sort64(uint64_t key, int lane) {
#pragma unroll
  for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
      const uint64_t other = __shfl_xor(key, j, 64);
      const bool up = (lane & k) == 0;
      const bool lower = (lane & j) == 0;
      const uint64_t mn = key < other ? key : other;
      const uint64_t mx = key < other ? other : key;
      key = (lower == up) ? mn : mx;
    }
  }
  return key;
}

// input: bitonic sequence across lanes; output ascending
__device__ __forceinline__ uint64_t wave_bitonic_merge64(uint64_t key, int lane) {
#pragma unroll
  for (int j = 32; j > 0; j >>= 1) {
    const uint64_t other = __shfl_xor(key, j, 64);
    const uint64_t mn = key < other ? key : other;
    const uint64_t mx = key < other ? other : key;
    key = (lane & j) == 0 ? mn : mx;
  }
  return key;
}

}  // namespace

// the fp32 scan publishes two sorted key lists per (query, chunk): one per wave row of the 8-wave kernel (k_flat8.hip)
uint32_t scan_lists_per_chunk() { return 2u; }

hipError_t launch_flat_scan(const ScanArgs& a, hipStream_t st) { return launch_flat_scan8(a, st); }

// ---------------------------------------------------------------------------------------------
// merge of the per-chunk sorted key lists: one wave per query
// ---------------------------------------------------------------------------------------------
// seed != 0: `merged` already holds the 64 best keys of earlier passes and is merged with the new lists.
// gthr != nullptr: also publish the k'-th best key so far (an upper bound of the query's final k'-th
// best) as the threshold the next scan pass starts from.
__global__ __launch_bounds__(64) void flat_merge_kernel(const uint64_t* __restrict__ part, uint32_t n_chunks,
                                                        uint32_t kprime, uint64_t* __restrict__ merged,
                                                        uint32_t lists_stride, uint32_t seed,
                                                        unsigned long long* __restrict__ gthr) {
  const int lane = threadIdx.x;
  const uint32_t q = blockIdx.x;
  const uint64_t* p = part + (size_t)q * lists_stride * kprime;
  uint64_t best = seed ? merged[(size_t)q * 64 + lane] : kKeyInf;
  // Most lists have nothing to contribute (empty — all-INF — in the later passes of the cascade, or
  // entirely above the current 64th best): look at 64 list heads at a time, one per lane, and visit only
  // the lists whose head beats the 64th best so far.
  for (uint32_t base = 0; base < n_chunks; base += 64) {
    const uint32_t c_l = base + lane;
    const uint64_t head = c_l < n_chunks ? p[(size_t)c_l * kprime] : kKeyInf;
    uint64_t todo = __ballot(head < __shfl(best, 63, 64));
    while (todo) {
      const uint32_t c = base + (uint32_t)__builtin_ctzll(todo);
      todo &= todo - 1;
      const uint64_t v = lane < (int)kprime ? p[(size_t)c * kprime + lane] : kKeyInf;  // ascending
      if (__shfl(v, 0, 64) >= __shfl(best, 63, 64)) continue;  // (the bar has risen since the ballot)
      const uint64_t rv = __shfl(v, 63 - lane, 64);                                     // descending
      const uint64_t m = best < rv ? best : rv;  // the 64 smallest of the union, bitonic
      best = wave_bitonic_merge64(m, lane);
    }
  }
  merged[(size_t)q * 64 + lane] = best;
  if (gthr && lane == (int)kprime - 1) gthr[q] = best;
}

// sample pass of the fp16 filter scan: scores[row][q] of the first n_rows rows -> gthr[q] = the kprime-th
// smallest score (as a key with the largest id, so ties with it still pass the scan's `key < threshold`).
// It is the kprime-th best of a subset of the rows, hence an upper bound of the query's final kprime-th best.
__global__ __launch_bounds__(64) void sample_select_kernel(const float* __restrict__ scores, uint32_t n_rows,
                                                           uint32_t q_rows, uint32_t kprime,
                                                           unsigned long long* __restrict__ gthr) {
  const int lane = threadIdx.x;
  const uint32_t q = blockIdx.x;
  uint64_t best = kKeyInf;  // ascending best-64 so far
  for (uint32_t r0 = 0; r0 < n_rows; r0 += 64) {
    const uint32_t r = r0 + lane;
    uint64_t key = kKeyInf;
    if (r < n_rows) {
      const float sc = scores[(size_t)r * q_rows + q];
      if (sc == sc) key = ((uint64_t)f32_to_ordered(sc) << 32) | 0xFFFFFFFFull;
    }
    key = wave_sort64(key, lane);
    const uint64_t rv = __shfl(key, 63 - lane, 64);
    const uint64_t m = best < rv ? best : rv;
    best = wave_bitonic_merge64(m, lane);
  }
  if (lane == (int)kprime - 1) gthr[q] = best;
}

hipError_t launch_sample_select(const float* scores, uint32_t n_rows, uint32_t q_rows, uint32_t nq, uint32_t kprime,
                                unsigned long long* gthr, hipStream_t st) {
  hipLaunchKernelGGL(sample_select_kernel, dim3(nq), dim3(64), 0, st, scores, n_rows, q_rows, kprime, gthr);
  return hipGetLastError();
}

// after the sample pass: the k'-th best key of the merged sample lists is an upper bound of the
// query's global k'-th best -> initial threshold of the main pass
__global__ __launch_bounds__(256) void set_gthr_kernel(const uint64_t* __restrict__ merged, uint32_t nq, uint32_t kprime,
                                                       unsigned long long* __restrict__ gthr) {
  const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q < nq) gthr[q] = merged[(size_t)q * 64 + kprime - 1];
}

hipError_t launch_set_gthr(const uint64_t* merged, uint32_t nq, uint32_t kprime, unsigned long long* gthr, hipStream_t st) {
  hipLaunchKernelGGL(set_gthr_kernel, dim3((nq + 255) / 256), dim3(256), 0, st, merged, nq, kprime, gthr);
  return hipGetLastError();
}

hipError_t launch_flat_merge(const uint64_t* part, uint32_t nq, uint32_t n_chunks, uint32_t kprime,
                             uint64_t* merged, hipStream_t st, uint32_t lists_stride, bool seed,
                             unsigned long long* gthr) {
  hipLaunchKernelGGL(flat_merge_kernel, dim3(nq), dim3(64), 0, st, part, n_chunks, kprime, merged, lists_stride,
                     seed ? 1u : 0u, gthr);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// canonical re-rank.  Distances are recomputed in exactly the order of hnswlib's SSE kernels
// (space_l2.h / space_ip.h; dispatch in L2Space / InnerProductSpace constructors), which is what
// oracle/hnsw_oracle.hpp restates: 4 strided partial sums over the multiple-of-4 body (multiply
// and add NOT fused), horizontal sum t0+t1+t2+t3 left to right, scalar tail added afterwards.
// One 4-lane group per candidate; lane j of the group plays SSE lane j.
// ---------------------------------------------------------------------------------------------
template <typename XT>
__global__ __launch_bounds__(256) void rerank_kernel(const RerankArgs a) {
  __shared__ uint64_t keys[64];
  __shared__ float approx[64];
  const int tid = threadIdx.x;
  const uint32_t q = blockIdx.x;
  const int g = tid >> 2, sub = tid & 3;
  const uint64_t mk = a.merged[(size_t)q * 64 + g];
  const uint32_t id = (uint32_t)mk;
  const bool valid = (g < (int)a.kprime) && (mk != kKeyInf) && (id < a.n);
  float d = __builtin_inff();
  if (valid) {
    const float* qv = a.Q + (size_t)q * a.ld;
    const XT* xv = (const XT*)a.X + (size_t)id * a.ld;
    const bool scale_x = a.metric == 2;
    const float xs = scale_x ? a.inv_norm[id] : 1.0f;
    d = canon_dist(a.metric == 0 ? 0 : 1, qv, xv, xs, scale_x, a.dims, sub);
  }
  if (sub == 0) {
    // (a NaN distance — a row or query holding NaN — is never a neighbour: the key is dropped)
    keys[g] = (valid && d == d) ? (((uint64_t)f32_to_ordered(d) << 32) | id) : kKeyInf;
    approx[g] = valid ? ordered_to_f32((uint32_t)(mk >> 32)) : __builtin_inff();
  }
  __syncthreads();
  if (tid < 64) {
    uint64_t key = wave_sort64(keys[tid], tid);
    const uint64_t nvalid_mask = __ballot(key != kKeyInf);
    const uint32_t nvalid = __builtin_popcountll(nvalid_mask);
    const uint32_t cnt = nvalid < a.k ? nvalid : a.k;
    // results go to columns [out_offset, out_offset + k) of a row of out_stride entries (paged large-k
    // requests write one page per call; out_stride == 0: the plain [nq][k] layout)
    const size_t ostride = a.out_stride ? a.out_stride : a.k;
    if (tid < (int)a.k) {
      const bool ok = (uint32_t)tid < cnt;
      a.out_ids[(size_t)q * ostride + a.out_offset + tid] = ok ? (uint64_t)(uint32_t)key : ~0ull;
      a.out_dist[(size_t)q * ostride + a.out_offset + tid] = ok ? ordered_to_f32((uint32_t)(key >> 32)) : __builtin_inff();
    }
    if (tid == 0) a.out_count[q] = (a.out_offset ? a.out_count[q] : 0u) + cnt;
    // certification: every row that is NOT a candidate has approx score >= the worst candidate's
    // approx score A_last (for L2 the scan's score omits |q|^2, added back here).  If
    // A_last - margin > exact k-th distance, no outsider can beat the k-th result, so the top-k is
    // provably the exhaustive top-k.  margin (cert_margin, ehx_kernels.h) is a worst-case bound of the gap
    // between an outsider's scan score and its canonical-order distance.  (Skipped when every row is a
    // candidate.)
    bool uncert = false;
    if (a.exact_keys) {
      // the keys are canonical distances of every row (exhaustive pass): nothing to certify
    } else if (a.n > a.kprime && cnt == a.k && a.k > 0) {
      float worst = -__builtin_inff();
      for (int j = 0; j < 64; ++j)
        if (approx[j] != __builtin_inff() && approx[j] > worst) worst = approx[j];
      float qn = 1.0f;  // |q|^2 of the query the distances are taken from (cosine: the normalised query)
      if (a.quv) {
        // filter keys: `worst` is a lower bound S of every outsider's score; map it to a distance
        // (NaN u marks a query the filter could not bound: the comparison below fails)
        const float2 uv = a.quv[q];
        worst = __builtin_fmaf(uv.x, worst, uv.y);
        qn = a.metric == 0 ? uv.y : (a.metric == 1 ? uv.x * uv.x : 1.0f);
      } else {
        const float* qv = a.Q + (size_t)q * a.ld;
        qn = 0.0f;
        for (uint32_t m = tid; m < a.dims; m += 64) qn += qv[m] * qv[m];
        for (int o = 32; o > 0; o >>= 1) qn += __shfl_xor(qn, o, 64);
        if (a.metric == 0) worst += qn;
      }
      const float kth = ordered_to_f32((uint32_t)(__shfl(key, (int)a.k - 1, 64) >> 32));
      const float margin = cert_margin(a.metric, a.dims, qn, a.max_sumsq ? *a.max_sumsq : __builtin_inff(),
                                       fmaxf(fabsf(kth), fabsf(worst)));
      uncert = !(worst - margin > kth);
    } else if (a.quv && a.n > a.kprime && cnt < a.k) {
      uncert = true;  // the filter lost candidates (overflowing gamma etc.): let the fp32 scan decide
    }
    if (tid == 0) {
      if (uncert) atomicAdd(a.n_uncertified, 1ull);
      if (a.uncert_flags) a.uncert_flags[q] = uncert ? 1u : 0u;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Last resort for a query whose top-k not even the fp32 scan can certify (near-ties finer than the
// fp32 rounding of the scan's arithmetic): the canonical distance of EVERY row, exactly as the re-rank
// computes it, best 64 keys (distance, id) per block of `rows_per_block` rows.  HBM-bound and slow
// (the whole shard per query) — the engine runs it for a handful of queries per batch at most.
// Grid (n_blocks, n_queries); the keys are exact, so their merge + re-rank needs no certification.
// ---------------------------------------------------------------------------------------------
template <typename XT>
__global__ __launch_bounds__(256) void exhaustive_kernel(const float* __restrict__ Q, const XT* __restrict__ X,
                                                         const float* __restrict__ inv_norm, uint32_t n, uint32_t dims,
                                                         uint32_t ld, int metric, uint32_t rows_per_block,
                                                         const uint64_t* __restrict__ floor,
                                                         uint64_t* __restrict__ out) {
  __shared__ uint64_t keys[64];
  const int tid = threadIdx.x;
  const uint32_t b = blockIdx.x, j = blockIdx.y;
  const int g = tid >> 2, sub = tid & 3;
  const float* qv = Q + (size_t)j * ld;
  const bool scale_x = metric == 2;
  const uint32_t r0 = b * rows_per_block;
  const uint32_t r1 = r0 + rows_per_block < n ? r0 + rows_per_block : n;
  // paging (k > 64): only keys strictly above the last key of the previous page count
  const uint64_t fl = floor ? floor[j] : 0ull;
  const bool paged = floor != nullptr;
  uint64_t best = kKeyInf;
  for (uint32_t base = r0; base < r1; base += 64) {
    const uint32_t id = base + (uint32_t)g;
    float d = __builtin_inff();
    if (id < r1) {
      const float xs = scale_x ? inv_norm[id] : 1.0f;
      d = canon_dist(metric == 0 ? 0 : 1, qv, X + (size_t)id * ld, xs, scale_x, dims, sub);
    }
    if (sub == 0) {
      uint64_t key = (id < r1 && d == d) ? (((uint64_t)f32_to_ordered(d) << 32) | id) : kKeyInf;  // NaN: not a neighbour
      if (paged && key <= fl) key = kKeyInf;
      keys[g] = key;
    }
    __syncthreads();
    if (tid < 64) {
      const uint64_t key = wave_sort64(keys[tid], tid);
      const uint64_t rv = __shfl(key, 63 - tid, 64);
      const uint64_t m = best < rv ? best : rv;
      best = wave_bitonic_merge64(m, tid);
    }
    __syncthreads();
  }
  if (tid < 64) out[((size_t)j * gridDim.x + b) * 64 + tid] = best;
}

// ---------------------------------------------------------------------------------------------
// ONE query from host memory against a small shard, in ONE launch (round 4; BASELINE configs[0], the reference's own
// request shape: one NearestNeighbor RPC = one query, server.cc:172-210).  The three-launch exhaustive path took
// ~130 us per ehx_knn call on 10 000 x 128 rows — two staged copies, three launches and a stream synchronisation, for
// 5 MB of rows.  Here the query is read straight from host-visible pinned memory, every workgroup prepares it itself
// (cosine: the canonical sequential-sum norm), computes the canonical distance of its rows and publishes its 64 best
// keys; the LAST workgroup to finish (a ticket) merges the lists and writes ids, distances and the count straight into
// host-visible memory, then raises a flag the host is spinning on: no copy, no synchronisation call.
// ---------------------------------------------------------------------------------------------
template <typename XT>
__global__ __launch_bounds__(256) void single_query_kernel(const SingleQueryArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem1[];
  float* qs = (float*)smem1;                       // [ld] prepared query
  float* sq = qs + a.ld;                           // [ld] squares (cosine norm)
  __shared__ uint64_t keys[64];
  __shared__ uint64_t wbest[4][64];
  __shared__ uint32_t last_s;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const uint32_t b = blockIdx.x;
  for (uint32_t i = tid; i < a.ld; i += 256) {
    const float v = i < a.dims ? a.q_in[i] : 0.0f;
    qs[i] = v;
    sq[i] = ex_mul(v, v);
  }
  __syncthreads();
  if (a.metric == 2) {
    if (tid == 0) {  // ONE sequential sum, hnswlib-python's order (prep_query_row's arithmetic)
      const float sum = seq_sum_lds(sq, a.dims);
      sq[0] = ex_div(1.0f, ex_add(ex_sqrt(sum), 1e-30f));
    }
    __syncthreads();
    const float inv = sq[0];
    __syncthreads();
    for (uint32_t i = tid; i < a.dims; i += 256) qs[i] = ex_mul(qs[i], inv);
    __syncthreads();
  }
  const XT* X = (const XT*)a.X;
  const int g = tid >> 2, sub = tid & 3;
  const bool scale_x = a.metric == 2;
  const uint32_t r0 = b * a.rows_per_block;
  const uint32_t r1 = r0 + a.rows_per_block < a.n ? r0 + a.rows_per_block : a.n;
  uint64_t best = kKeyInf;
  for (uint32_t base = r0; base < r1; base += 64) {
    const uint32_t id = base + (uint32_t)g;
    float d = __builtin_inff();
    if (id < r1) {
      const float xs = scale_x ? a.inv_norm[id] : 1.0f;
      d = canon_dist(a.metric == 0 ? 0 : 1, qs, X + (size_t)id * a.ld, xs, scale_x, a.dims, sub);
    }
    if (sub == 0) keys[g] = (id < r1 && d == d) ? (((uint64_t)f32_to_ordered(d) << 32) | id) : kKeyInf;  // NaN: no neighbour
    __syncthreads();
    if (tid < 64) {
      const uint64_t key = wave_sort64(keys[tid], tid);
      const uint64_t rv = __shfl(key, 63 - tid, 64);
      const uint64_t m = best < rv ? best : rv;
      best = wave_bitonic_merge64(m, tid);
    }
    __syncthreads();
  }
  if (tid < 64) a.part[(size_t)b * 64 + tid] = best;
  __threadfence();   // this workgroup's list is visible device-wide before its ticket is
  __syncthreads();
  if (tid == 0) {
    const uint32_t t = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    last_s = t == gridDim.x - 1 ? 1u : 0u;
  }
  __syncthreads();
  if (!last_s) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // the other workgroups' lists (other XCDs' L2s)
  // the last workgroup merges the gridDim.x lists: wave w takes lists w, w + 4, ...; then wave 0 the four results
  uint64_t mine = kKeyInf;
  for (uint32_t c = (uint32_t)wv; c < gridDim.x; c += 4) {
    const uint64_t v = __hip_atomic_load(a.part + (size_t)c * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (__shfl(v, 0, 64) >= __shfl(mine, 63, 64)) continue;
    const uint64_t rv = __shfl(v, 63 - lane, 64);
    const uint64_t m = mine < rv ? mine : rv;
    mine = wave_bitonic_merge64(m, lane);
  }
  wbest[wv][lane] = mine;
  __syncthreads();
  if (wv == 0) {
    uint64_t fin = wbest[0][lane];
#pragma unroll
    for (int w2 = 1; w2 < 4; ++w2) {
      const uint64_t v = wbest[w2][lane];
      const uint64_t rv = __shfl(v, 63 - lane, 64);
      const uint64_t m = fin < rv ? fin : rv;
      fin = wave_bitonic_merge64(m, lane);
    }
    const uint32_t nvalid = (uint32_t)__builtin_popcountll(__ballot(fin != kKeyInf));
    const uint32_t cnt = nvalid < a.k ? nvalid : a.k;
    if ((uint32_t)lane < a.k) {
      const bool ok = (uint32_t)lane < cnt;
      a.out_ids[lane] = ok ? (uint64_t)(uint32_t)fin : ~0ull;
      a.out_dist[lane] = ok ? ordered_to_f32((uint32_t)(fin >> 32)) : __builtin_inff();
    }
    if (lane == 0) {
      a.out_count[0] = cnt;
      __hip_atomic_store(a.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next call
    }
    __threadfence_system();
    __builtin_amdgcn_s_waitcnt(0);
    if (lane == 0) __hip_atomic_store(a.done_flag, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

hipError_t launch_single_query(const SingleQueryArgs& a, uint32_t n_blocks, hipStream_t st) {
  const size_t lds = (size_t)a.ld * 8;
  static DynLdsAttr attr;
  const void* fns[2] = {(const void*)single_query_kernel<float>, (const void*)single_query_kernel<__half>};
  if (hipError_t e = attr.ensure(fns, 2, lds + 4096); e != hipSuccess) return e;
  if (a.x_half) hipLaunchKernelGGL(single_query_kernel<__half>, dim3(n_blocks), dim3(256), lds, st, a);
  else hipLaunchKernelGGL(single_query_kernel<float>, dim3(n_blocks), dim3(256), lds, st, a);
  return hipGetLastError();
}

// next page's floor = the 64th (last) key of this page; exhausted queries get INF (nothing above it)
__global__ __launch_bounds__(256) void set_floor_kernel(const uint64_t* __restrict__ merged, uint32_t nq,
                                                        uint64_t* __restrict__ floor) {
  const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q < nq) floor[q] = merged[(size_t)q * 64 + 63];
}

hipError_t launch_set_floor(const uint64_t* merged, uint32_t nq, uint64_t* floor, hipStream_t st) {
  hipLaunchKernelGGL(set_floor_kernel, dim3((nq + 255) / 256), dim3(256), 0, st, merged, nq, floor);
  return hipGetLastError();
}

hipError_t launch_exhaustive(const float* Q, const void* X, int x_half, const float* inv_norm, uint32_t n, uint32_t dims,
                             uint32_t ld, int metric, uint32_t rows_per_block, uint32_t n_blocks, uint32_t nq,
                             const uint64_t* floor, uint64_t* out, hipStream_t st) {
  const dim3 grid(n_blocks, nq);
  if (x_half)
    hipLaunchKernelGGL(exhaustive_kernel<__half>, grid, dim3(256), 0, st, Q, (const __half*)X, inv_norm, n, dims, ld,
                       metric, rows_per_block, floor, out);
  else
    hipLaunchKernelGGL(exhaustive_kernel<float>, grid, dim3(256), 0, st, Q, (const float*)X, inv_norm, n, dims, ld,
                       metric, rows_per_block, floor, out);
  return hipGetLastError();
}

hipError_t launch_rerank(const RerankArgs& a, hipStream_t st) {
  if (a.x_half) hipLaunchKernelGGL(rerank_kernel<__half>, dim3(a.nq), dim3(256), 0, st, a);
  else hipLaunchKernelGGL(rerank_kernel<float>, dim3(a.nq), dim3(256), 0, st, a);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// k-way merge of per-shard result lists (after the RCCL all-gather): one wave per query.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void merge_lists_kernel(const uint64_t* __restrict__ ids,
                                                         const float* __restrict__ dist,
                                                         const uint32_t* __restrict__ count, uint32_t nq,
                                                         uint32_t k, uint32_t n_lists, size_t ids_stride,
                                                         size_t dist_stride, size_t count_stride,
                                                         uint64_t id_mul, uint64_t id_step,
                                                         uint64_t* __restrict__ out_ids,
                                                         float* __restrict__ out_dist,
                                                         uint32_t* __restrict__ out_count) {
  // lists are sorted nearest-first; k <= 64.  Keys are (ordered dist, list, pos) so the wave
  // sort is stable w.r.t. (dist, id) once ties are broken by id below.
  const int lane = threadIdx.x;
  const uint32_t q = blockIdx.x;
  // running best: (dist, id) pairs, one per lane, sorted by (dist, id)
  float bd = __builtin_inff();
  uint64_t bi = ~0ull;
  uint32_t total = 0;
  for (uint32_t l = 0; l < n_lists; ++l) {
    // list l of each array starts l * stride BYTES after list 0 (natural layout or one packed gather buffer)
    const uint64_t* il = (const uint64_t*)((const char*)ids + l * ids_stride) + (size_t)q * k;
    const float* dl = (const float*)((const char*)dist + l * dist_stride) + (size_t)q * k;
    const uint32_t c = count ? ((const uint32_t*)((const char*)count + l * count_stride))[q] : k;
    total += c;
    float d = (lane < (int)k && (uint32_t)lane < c) ? dl[lane] : __builtin_inff();
    uint64_t i = (lane < (int)k && (uint32_t)lane < c) ? il[lane] * id_mul + (uint64_t)l * id_step : ~0ull;
    // reverse incoming list, elementwise min by (dist, id), bitonic merge on the pair
    const float rd = __shfl(d, 63 - lane, 64);
    const uint64_t ri = __shfl(i, 63 - lane, 64);
    const bool take = (rd < bd) || (rd == bd && ri < bi);
    if (take) {
      bd = rd;
      bi = ri;
    }
#pragma unroll
    for (int j = 32; j > 0; j >>= 1) {
      const float od = __shfl_xor(bd, j, 64);
      const uint64_t oi = __shfl_xor(bi, j, 64);
      const bool less = (od < bd) || (od == bd && oi < bi);  // other < mine
      const bool want_min = (lane & j) == 0;
      if (want_min == less) {
        bd = od;
        bi = oi;
      }
    }
  }
  const uint32_t cnt = total < k ? total : k;
  if (lane < (int)k) {
    const bool ok = (uint32_t)lane < cnt;
    out_ids[(size_t)q * k + lane] = ok ? bi : ~0ull;
    out_dist[(size_t)q * k + lane] = ok ? bd : __builtin_inff();
  }
  if (lane == 0 && out_count) out_count[q] = cnt;
}

// the same merge for k > 64 (sharded spaces serve k up to 1024 like unsharded ones): lane l walks list l, every step
// the wave takes the smallest head by (distance, id) and the lane that held it moves on — k steps of one load and a
// six-stage minimum; n_lists <= 64
__global__ __launch_bounds__(64) void merge_lists_walk_kernel(const uint64_t* __restrict__ ids,
                                                              const float* __restrict__ dist,
                                                              const uint32_t* __restrict__ count, uint32_t nq,
                                                              uint32_t k, uint32_t n_lists, size_t ids_stride,
                                                              size_t dist_stride, size_t count_stride,
                                                              uint64_t id_mul, uint64_t id_step,
                                                              uint64_t* __restrict__ out_ids,
                                                              float* __restrict__ out_dist,
                                                              uint32_t* __restrict__ out_count) {
  const int lane = threadIdx.x;
  const uint32_t q = blockIdx.x;
  const bool mine = (uint32_t)lane < n_lists;
  const uint64_t* il = (const uint64_t*)((const char*)ids + (size_t)lane * ids_stride) + (size_t)q * k;
  const float* dl = (const float*)((const char*)dist + (size_t)lane * dist_stride) + (size_t)q * k;
  uint32_t c = 0;
  if (mine) c = count ? ((const uint32_t*)((const char*)count + (size_t)lane * count_stride))[q] : k;
  if (c > k) c = k;
  uint32_t total = c;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) total += __shfl_xor(total, o, 64);
  const uint32_t cnt = total < k ? total : k;
  uint32_t pos = 0;
  float hd = pos < c ? dl[pos] : __builtin_inff();
  uint64_t hi = pos < c ? il[pos] * id_mul + (uint64_t)lane * id_step : ~0ull;
  for (uint32_t j = 0; j < k; ++j) {
    float md = hd;
    uint64_t mi = hi;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const float od = __shfl_xor(md, o, 64);
      const uint64_t oi = __shfl_xor(mi, o, 64);
      if (od < md || (od == md && oi < mi)) {
        md = od;
        mi = oi;
      }
    }
    if (lane == 0) {
      out_ids[(size_t)q * k + j] = j < cnt ? mi : ~0ull;
      out_dist[(size_t)q * k + j] = j < cnt ? md : __builtin_inff();
    }
    if (hi == mi && hi != ~0ull) {  // (global ids are distinct: exactly one lane holds the winner)
      pos += 1;
      hd = pos < c ? dl[pos] : __builtin_inff();
      hi = pos < c ? il[pos] * id_mul + (uint64_t)lane * id_step : ~0ull;
    }
  }
  if (lane == 0 && out_count) out_count[q] = cnt;
}

hipError_t launch_merge_lists(const uint64_t* ids, const float* dist, const uint32_t* count, uint32_t nq,
                              uint32_t k, uint32_t n_lists, uint64_t* out_ids, float* out_dist,
                              uint32_t* out_count, hipStream_t st, size_t ids_stride, size_t dist_stride,
                              size_t count_stride, uint64_t id_mul, uint64_t id_step) {
  if (k > 64) {
    if (n_lists > 64) return hipErrorInvalidValue;
    hipLaunchKernelGGL(merge_lists_walk_kernel, dim3(nq), dim3(64), 0, st, ids, dist, count, nq, k, n_lists, ids_stride,
                       dist_stride, count_stride, id_mul, id_step, out_ids, out_dist, out_count);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(merge_lists_kernel, dim3(nq), dim3(64), 0, st, ids, dist, count, nq, k, n_lists, ids_stride,
                     dist_stride, count_stride, id_mul, id_step, out_ids, out_dist, out_count);
  return hipGetLastError();
}

}  // namespace ehx
