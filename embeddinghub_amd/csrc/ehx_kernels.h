// Internal launch interface between the C-ABI host code (ehx_*.cpp) and the gfx950 kernels.
// Not part of the public boundary (include/ehx.h is).
#pragma once

#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <atomic>
#include <stdint.h>

namespace ehx {

constexpr uint32_t kTileRows = 128;   // corpus rows per scan tile
constexpr uint32_t kTileQ = 256;      // queries per scan tile
constexpr uint32_t kTileRows16 = 256; // corpus rows per tile of the fp16 filter scan (k_flat16.hip)
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
// LDS hand-over between the lanes of ONE wave (kernels launched with 64 threads per workgroup): the LDS accesses of
// a wave execute in program order, so a compiler-level fence is all a write -> read across lanes needs — no
// s_barrier and no drain of the LDS queue, which __syncthreads() would put on the dependent chain
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// A value that is the same in every lane but reaches the wave through a shuffle or an LDS read is "divergent" to the
// compiler: everything derived from it sits in vector registers behind exec-mask branches.  readfirstlane moves it
// to a scalar register, and loop bookkeeping built on it runs on the scalar unit.
__device__ __forceinline__ uint32_t wave_uniform(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }
__device__ __forceinline__ float wave_uniform(float x) {
  return __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(x)));
}

__device__ __forceinline__ float ex_add(float a, float b) { return a + b; }
__device__ __forceinline__ float ex_sub(float a, float b) { return a - b; }
__device__ __forceinline__ float ex_mul(float a, float b) { return a * b; }
__device__ __forceinline__ float ex_div(float a, float b) { return a / b; }
__device__ __forceinline__ float ex_sqrt(float a) { return __builtin_sqrtf(a); }

// Certification margin of the re-rank (rerank_kernel*): an upper bound of
//     | scan score of a row  -  that row's canonical (oracle-order) distance |
// for EVERY row of the space, so that  "worst kept score - margin > exact k-th distance"  proves that no
// row outside the candidate list can enter the top-k.  Both numbers are fp32 evaluations of the same real
// quantity T over the same fp32 inputs; with u = 2^-24 and Higham's gamma_n ~ n*u:
//   * matrix-core scan: one fma chain over d products                      -> |err| <= gamma_d     * W
//   * canonical order : products rounded, 4 chains of d/4 adds, 3 joining adds, cosine also rounds
//     x_i * inv_norm                                                        -> |err| <= gamma_(d/4+5) * W
//   with W = sum|q_i x_i| <= |q| |x| (Cauchy-Schwarz): cosine W <= 1.01 (both operands normalised by their
//   fp32 norms), inner product W <= |q| * max|x|, L2^2: every term is bounded by (|q| + max|x|)^2 (the row and
//   query norms enter the scan score through their own d-term sums).
//   eps_d = 1.3 * (d + 16) * u  >=  gamma_d + gamma_(d/4+5)  for every d <= 65536.
//   * the last affine operations (score = dot*a + b, D = u*S + v, 1 - sum) round relative to the result:
//     2e-6 * scale  (>= 32 u).
// Filter scans (fp16 / int8 lower bounds) need only the canonical-order term; the same formula covers them.
// max_sumsq = the largest |x|^2 ever written to the space (launch_row_stats); Inf or NaN there makes the
// margin infinite, i.e. nothing is certified and the exhaustive canonical pass answers.
__device__ __forceinline__ float cert_margin(int metric, uint32_t dims, float qn, float max_sumsq, float scale) {
  const float eps_d = 1.3f * ((float)dims + 16.0f) * 5.9604645e-8f;
  float base;
  if (metric == 2) {
    base = eps_d * 1.01f;
  } else {
    const float qb = __builtin_sqrtf(qn), mx = __builtin_sqrtf(max_sumsq);
    base = metric == 1 ? eps_d * 1.01f * qb * mx : eps_d * 1.01f * (qb + mx) * (qb + mx);
  }
  return base + 2e-6f * fmaxf(scale, fmaxf(qn, 1.0f));
}

// Canonical distance between a prepared query and a stored row, in exactly the order of hnswlib's SSE
// kernels (space_l2.h / space_ip.h; oracle/hnsw_oracle.hpp restates them): 4 strided partial sums
// over the multiple-of-4 body (multiply and add NOT fused), horizontal sum t0+t1+t2+t3 left to
// right, scalar tail added afterwards.  Executed by a 4-lane group: lane `sub` plays SSE lane `sub`;
// all 4 lanes of the group must be active.  metric01: 0 = L2^2, 1 = 1 - inner product.  For cosine
// the stored row is normalised on the fly (x * inv_norm, one rounding — hnswlib-python's
// normalize_vector) and the query arrives normalised.  Result valid in sub-lane 0.
// row element load: fp32 rows as stored, fp16 rows widened exactly (every half is a float)
__device__ __forceinline__ float ld_row(const float* x, uint32_t m) { return x[m]; }
__device__ __forceinline__ float ld_row(const __half* x, uint32_t m) { return __half2float(x[m]); }

template <typename XT>
__device__ __forceinline__ float canon_dist(int metric, const float* __restrict__ q,
                                            const XT* __restrict__ x, float xscale, bool scale_x,
                                            uint32_t dims, int sub) {
  uint32_t body;
  if ((dims & 15u) == 0 || (dims & 3u) == 0) body = dims;
  else if (dims > 16) body = dims & ~15u;
  else if (dims > 4) body = dims & ~3u;
  else body = 0;
  float part = 0.0f;
  if (metric == 0) {
    for (uint32_t m = sub; m < body; m += 4) {
      const float xv = scale_x ? ex_mul(ld_row(x, m), xscale) : ld_row(x, m);
      const float diff = ex_sub(q[m], xv);
      part = ex_add(part, ex_mul(diff, diff));
    }
  } else {
    for (uint32_t m = sub; m < body; m += 4) {
      const float xv = scale_x ? ex_mul(ld_row(x, m), xscale) : ld_row(x, m);
      part = ex_add(part, ex_mul(q[m], xv));
    }
  }
  // horizontal sum in lane order within the 4-lane group
  const float t1 = __shfl_down(part, 1, 4), t2 = __shfl_down(part, 2, 4), t3 = __shfl_down(part, 3, 4);
  float res = ex_add(ex_add(ex_add(part, t1), t2), t3);
  if (body != dims) {
    float tail = 0.0f;
    if (metric == 0) {
      for (uint32_t m = body; m < dims; ++m) {
        const float xv = scale_x ? ex_mul(ld_row(x, m), xscale) : ld_row(x, m);
        const float diff = ex_sub(q[m], xv);
        tail = ex_add(tail, ex_mul(diff, diff));
      }
    } else {
      for (uint32_t m = body; m < dims; ++m) {
        const float xv = scale_x ? ex_mul(ld_row(x, m), xscale) : ld_row(x, m);
        tail = ex_add(tail, ex_mul(q[m], xv));
      }
    }
    if (metric != 0 && body) {
      // hnswlib 0.5.x residual variants of the inner product: both halves are already distances (1 - sum) and are
      // combined as  res + res_tail - 1.0f  (space_ip.h; oracle/hnsw_oracle.hpp:ip_dist)
      return ex_sub(ex_add(ex_sub(1.0f, res), ex_sub(1.0f, tail)), 1.0f);
    }
    res = body ? ex_add(res, tail) : tail;
  }
  if (metric != 0) res = ex_sub(1.0f, res);
  return res;  // valid in sub-lane 0
}

// Same canonical arithmetic executed by ONE lane: the lane keeps the 4 SSE partial sums itself and
// walks its row with 16-byte loads (q may live in LDS).  Used by the graph search and the graph
// insertion, where every lane owns one neighbour row.  Requires 16-byte aligned q and x (row stride
// ld % 4 == 0).
//
// The walk is HBM-latency bound unless many loads are in flight per lane (a row is ld*4 contiguous
// bytes = ld/32 cache lines that nobody else touches; the plain loop compiles to ONE 16-byte load in
// flight per lane): the body is cut into blocks of kLaneBlk 16-byte loads and a ring of three
// register blocks keeps two blocks in flight ahead of the block being accumulated.  Measured on the
// graph bench (profiles/r01_m_*): 16-24 loads in flight per lane are enough — kLaneBlk 8 and 16 are
// within 3 % of each other, 4 is 6 % slower at d=768 — because past that point the random row gathers
// are bound by what the memory system delivers for this pattern (scripts/ubench/gather_rows.hip:
// 27 lanes x private rows, 4 waves per CU: 4.2-4.5 TB/s; deeper queues thrash the 32-KiB L1).
// The accumulation order is unchanged (block after block, 16 bytes after 16 bytes), so the result is
// bit-identical to the plain loop.
#ifndef EHX_LANE_BLK
#define EHX_LANE_BLK 8
#endif
constexpr int kLaneBlk = EHX_LANE_BLK;  // 16-byte loads per ring block (8: 128 B = one cache line per lane)

template <int METRIC01, bool SCALE>
__device__ __forceinline__ void canon_lane_step(float4 xv, const float4 qv, float xscale, float& p0, float& p1,
                                                float& p2, float& p3) {
  if (SCALE) {
    xv.x = ex_mul(xv.x, xscale);
    xv.y = ex_mul(xv.y, xscale);
    xv.z = ex_mul(xv.z, xscale);
    xv.w = ex_mul(xv.w, xscale);
  }
  if (METRIC01 == 0) {
    const float d0 = ex_sub(qv.x, xv.x), d1 = ex_sub(qv.y, xv.y), d2 = ex_sub(qv.z, xv.z), d3 = ex_sub(qv.w, xv.w);
    p0 = ex_add(p0, ex_mul(d0, d0));
    p1 = ex_add(p1, ex_mul(d1, d1));
    p2 = ex_add(p2, ex_mul(d2, d2));
    p3 = ex_add(p3, ex_mul(d3, d3));
  } else {
    p0 = ex_add(p0, ex_mul(qv.x, xv.x));
    p1 = ex_add(p1, ex_mul(qv.y, xv.y));
    p2 = ex_add(p2, ex_mul(qv.z, xv.z));
    p3 = ex_add(p3, ex_mul(qv.w, xv.w));
  }
}

template <int METRIC01, bool SCALE, bool RING = true>
__device__ __forceinline__ float canon_dist_lane_t(const float* __restrict__ q, const float* __restrict__ x,
                                                   float xscale, uint32_t dims) {
  uint32_t body;
  if ((dims & 15u) == 0 || (dims & 3u) == 0) body = dims;
  else if (dims > 16) body = dims & ~15u;
  else if (dims > 4) body = dims & ~3u;
  else body = 0;
  float p0 = 0.0f, p1 = 0.0f, p2 = 0.0f, p3 = 0.0f;
  constexpr int BL = kLaneBlk;
  const uint32_t nblk = RING ? body / (4u * BL) : 0u;  // !RING: few registers, four loads in flight
  float4 r0[BL], r1[BL], r2[BL];
  const float4* x4 = (const float4*)x;
  const float4* q4 = (const float4*)q;
#define EHX_LANE_LOAD(R, B)                                        \
  _Pragma("unroll") for (int i_ = 0; i_ < BL; ++i_) R[i_] = x4[(size_t)(B) * BL + i_];
#define EHX_LANE_ACC(R, B)                                         \
  _Pragma("unroll") for (int i_ = 0; i_ < BL; ++i_)                \
      canon_lane_step<METRIC01, SCALE>(R[i_], q4[(size_t)(B) * BL + i_], xscale, p0, p1, p2, p3);
  uint32_t b = 0;
  if (nblk >= 2) {
    EHX_LANE_LOAD(r0, 0)
    EHX_LANE_LOAD(r1, 1)
    // steady state: three blocks accumulated per trip, every load two blocks ahead of its use, no branches
    for (; b + 5 <= nblk; b += 3) {
      EHX_LANE_LOAD(r2, b + 2)
      EHX_LANE_ACC(r0, b)
      EHX_LANE_LOAD(r0, b + 3)
      EHX_LANE_ACC(r1, b + 1)
      EHX_LANE_LOAD(r1, b + 4)
      EHX_LANE_ACC(r2, b + 2)
    }
    const uint32_t rem = nblk - b;  // 2, 3 or 4 blocks left; r0 / r1 hold blocks b / b+1
    if (rem >= 3) { EHX_LANE_LOAD(r2, b + 2) }
    EHX_LANE_ACC(r0, b)
    if (rem == 4) { EHX_LANE_LOAD(r0, b + 3) }
    EHX_LANE_ACC(r1, b + 1)
    if (rem >= 3) { EHX_LANE_ACC(r2, b + 2) }
    if (rem == 4) { EHX_LANE_ACC(r0, b + 3) }
  } else if (nblk == 1) {
    EHX_LANE_LOAD(r0, 0)
    EHX_LANE_ACC(r0, 0)
  }
#undef EHX_LANE_LOAD
#undef EHX_LANE_ACC
  // the 16-byte pieces after the last full block (fewer than kLaneBlk), four loads at a time
  {
    uint32_t m = nblk * BL;
    const uint32_t m1 = body / 4u;
    for (; m + 4 <= m1; m += 4) {
      const float4 t0 = x4[m], t1 = x4[m + 1], t2 = x4[m + 2], t3 = x4[m + 3];
      canon_lane_step<METRIC01, SCALE>(t0, q4[m], xscale, p0, p1, p2, p3);
      canon_lane_step<METRIC01, SCALE>(t1, q4[m + 1], xscale, p0, p1, p2, p3);
      canon_lane_step<METRIC01, SCALE>(t2, q4[m + 2], xscale, p0, p1, p2, p3);
      canon_lane_step<METRIC01, SCALE>(t3, q4[m + 3], xscale, p0, p1, p2, p3);
    }
    for (; m < m1; ++m) canon_lane_step<METRIC01, SCALE>(x4[m], q4[m], xscale, p0, p1, p2, p3);
  }
  float res = ex_add(ex_add(ex_add(p0, p1), p2), p3);
  if (body != dims) {
    float tail = 0.0f;
    for (uint32_t m = body; m < dims; ++m) {
      const float xv = SCALE ? ex_mul(x[m], xscale) : x[m];
      if (METRIC01 == 0) {
        const float diff = ex_sub(q[m], xv);
        tail = ex_add(tail, ex_mul(diff, diff));
      } else {
        tail = ex_add(tail, ex_mul(q[m], xv));
      }
    }
    if (METRIC01 != 0 && body) return ex_sub(ex_add(ex_sub(1.0f, res), ex_sub(1.0f, tail)), 1.0f);  // see canon_dist
    res = body ? ex_add(res, tail) : tail;
  }
  if (METRIC01 != 0) res = ex_sub(1.0f, res);
  return res;
}

// The same canonical arithmetic executed by a 4-LANE GROUP over a row of the graph-mode SEARCH COPY
// (launch_make_search_copy: inside every 16-float block the four inputs of SSE partial sum j are 16
// contiguous bytes; cosine rows are stored normalised).  Lane `sub` of the group plays SSE lane `sub`:
// per block it loads ONE float4 — the group reads the block as one coalesced 64-byte piece — and adds
// its four products to its partial sum in order.  q is the query permuted the same way (LDS).  All four
// lanes of the group must be active; every lane returns the full result.  Same register ring as above.
// SCALE (round 4, single-copy graph spaces: the rows are stored raw and permuted, not normalised): every element of
// the row is multiplied by xs first — hnswlib-python's stored normalised row x * inv_norm, one rounding per element,
// formed on the fly; the products with the query then see exactly the values the normalised copy held.
// x * xs per element as two packed multiplies (v_pk_mul_f32: IEEE, one rounding per element like the scalar form)
__device__ __forceinline__ float4 scale_f4(float4 xv, float xs) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const f32x2 sc = {xs, xs};
  f32x2 lo = {xv.x, xv.y}, hi = {xv.z, xv.w};
  lo = lo * sc;
  hi = hi * sc;
  return make_float4(lo.x, lo.y, hi.x, hi.y);
}

template <int METRIC01, bool SCALE = false>
__device__ __forceinline__ void canon_group_step(float4 xv, const float4 qv, float& p, int ncomp = 4, float xs = 1.0f) {
  if (SCALE) xv = scale_f4(xv, xs);
  if (METRIC01 == 0) {
    const float d0 = ex_sub(qv.x, xv.x), d1 = ex_sub(qv.y, xv.y), d2 = ex_sub(qv.z, xv.z), d3 = ex_sub(qv.w, xv.w);
    p = ex_add(p, ex_mul(d0, d0));
    if (ncomp > 1) p = ex_add(p, ex_mul(d1, d1));
    if (ncomp > 2) p = ex_add(p, ex_mul(d2, d2));
    if (ncomp > 3) p = ex_add(p, ex_mul(d3, d3));
  } else {
    p = ex_add(p, ex_mul(qv.x, xv.x));
    if (ncomp > 1) p = ex_add(p, ex_mul(qv.y, xv.y));
    if (ncomp > 2) p = ex_add(p, ex_mul(qv.z, xv.z));
    if (ncomp > 3) p = ex_add(p, ex_mul(qv.w, xv.w));
  }
}

// position of element m of a row inside the search copy / the permuted query
__host__ __device__ inline uint32_t search_copy_pos(uint32_t m) { return (m & ~15u) + ((m & 3u) << 2) + ((m >> 2) & 3u); }

template <int METRIC01, bool SCALE = false>
__device__ __forceinline__ float canon_dist_group_t(const float* __restrict__ qp, const float* __restrict__ xs, int sub,
                                                    uint32_t dims, float xscale = 1.0f) {
  uint32_t body;
  if ((dims & 15u) == 0 || (dims & 3u) == 0) body = dims;
  else if (dims > 16) body = dims & ~15u;
  else if (dims > 4) body = dims & ~3u;
  else body = 0;
  float p = 0.0f;
  constexpr int BL = kLaneBlk;
  const uint32_t n16 = body / 16u;       // full 16-float blocks: one float4 per lane each
  const uint32_t nblk = n16 / BL;        // ring blocks
  float4 r0[BL], r1[BL], r2[BL];
  const float4* x4 = (const float4*)xs + sub;  // block t of this lane: x4[4 t]
  const float4* q4 = (const float4*)qp + sub;
#define EHX_GRP_LOAD(R, B)                                        \
  _Pragma("unroll") for (int i_ = 0; i_ < BL; ++i_) R[i_] = x4[((size_t)(B) * BL + i_) * 4];
  // (SCALE: the ring block's query pieces are read from LDS together, ahead of its products — left to the scheduler,
  // the scaled walk reads one piece, waits for it, multiplies, and pays the LDS latency once per 16-float block)
#define EHX_GRP_ACC(R, B)                                                                                            \
  if (SCALE) {                                                                                                       \
    float4 qv_[BL];                                                                                                  \
    _Pragma("unroll") for (int i_ = 0; i_ < BL; ++i_) qv_[i_] = q4[((size_t)(B) * BL + i_) * 4];                      \
    _Pragma("unroll") for (int i_ = 0; i_ < BL; ++i_) R[i_] = scale_f4(R[i_], xscale);                               \
    _Pragma("unroll") for (int i_ = 0; i_ < BL; ++i_) canon_group_step<METRIC01, false>(R[i_], qv_[i_], p, 4, 1.0f); \
  } else {                                                                                                           \
    _Pragma("unroll") for (int i_ = 0; i_ < BL; ++i_)                                                                \
      canon_group_step<METRIC01, false>(R[i_], q4[((size_t)(B) * BL + i_) * 4], p, 4, 1.0f);                         \
  }
  uint32_t b = 0;
  if (nblk >= 2) {
    EHX_GRP_LOAD(r0, 0)
    EHX_GRP_LOAD(r1, 1)
    for (; b + 5 <= nblk; b += 3) {
      EHX_GRP_LOAD(r2, b + 2)
      EHX_GRP_ACC(r0, b)
      EHX_GRP_LOAD(r0, b + 3)
      EHX_GRP_ACC(r1, b + 1)
      EHX_GRP_LOAD(r1, b + 4)
      EHX_GRP_ACC(r2, b + 2)
    }
    const uint32_t rem = nblk - b;  // 2, 3 or 4 ring blocks left; r0 / r1 hold blocks b / b+1
    if (rem >= 3) { EHX_GRP_LOAD(r2, b + 2) }
    EHX_GRP_ACC(r0, b)
    if (rem == 4) { EHX_GRP_LOAD(r0, b + 3) }
    EHX_GRP_ACC(r1, b + 1)
    if (rem >= 3) { EHX_GRP_ACC(r2, b + 2) }
    if (rem == 4) { EHX_GRP_ACC(r0, b + 3) }
  } else if (nblk == 1) {
    EHX_GRP_LOAD(r0, 0)
    EHX_GRP_ACC(r0, 0)
  }
#undef EHX_GRP_LOAD
#undef EHX_GRP_ACC
  {
    uint32_t t = nblk * BL;  // 16-float blocks after the last full ring block (fewer than kLaneBlk), four at a time
    for (; t + 4 <= n16; t += 4) {
      const float4 t0 = x4[t * 4], t1 = x4[(t + 1) * 4], t2 = x4[(t + 2) * 4], t3 = x4[(t + 3) * 4];
      canon_group_step<METRIC01, SCALE>(t0, q4[t * 4], p, 4, xscale);
      canon_group_step<METRIC01, SCALE>(t1, q4[(t + 1) * 4], p, 4, xscale);
      canon_group_step<METRIC01, SCALE>(t2, q4[(t + 2) * 4], p, 4, xscale);
      canon_group_step<METRIC01, SCALE>(t3, q4[(t + 3) * 4], p, 4, xscale);
    }
    for (; t < n16; ++t) canon_group_step<METRIC01, SCALE>(x4[t * 4], q4[t * 4], p, 4, xscale);
    // the 4-float pieces of a last, partial block (body % 16 / 4 of them): components 0..rem4-1
    const int rem4 = (int)((body & 15u) >> 2);
    if (rem4) canon_group_step<METRIC01, SCALE>(x4[n16 * 4], q4[n16 * 4], p, rem4, xscale);
  }
  // horizontal sum in SSE-lane order: ((p0 + p1) + p2) + p3, formed by every lane of the group
  const float t0 = __shfl(p, 0, 4), t1 = __shfl(p, 1, 4), t2 = __shfl(p, 2, 4), t3 = __shfl(p, 3, 4);
  float res = ex_add(ex_add(ex_add(t0, t1), t2), t3);
  if (body != dims) {
    float tail = 0.0f;
    for (uint32_t m = body; m < dims; ++m) {
      const uint32_t pos = search_copy_pos(m);
      const float xv = SCALE ? ex_mul(xs[pos], xscale) : xs[pos];
      if (METRIC01 == 0) {
        const float diff = ex_sub(qp[pos], xv);
        tail = ex_add(tail, ex_mul(diff, diff));
      } else {
        tail = ex_add(tail, ex_mul(qp[pos], xv));
      }
    }
    if (METRIC01 != 0 && body) return ex_sub(ex_add(ex_sub(1.0f, res), ex_sub(1.0f, tail)), 1.0f);  // see canon_dist
    res = body ? ex_add(res, tail) : tail;
  }
  if (METRIC01 != 0) res = ex_sub(1.0f, res);
  return res;
}

// Two rows of exactly 16 * N16 floats by one 4-lane group, all 2 * N16 loads of the lane in flight before the first
// product (a 128-dim row is ONE ring block of canon_dist_group_t: with more than 16 fresh neighbours the second pass
// would wait a second memory round trip).  Same arithmetic and order per row as canon_dist_group_t.
template <int METRIC01, int N16, bool SCALE = false>
__device__ __forceinline__ void canon_dist_group_pair(const float* __restrict__ qp, const float* __restrict__ xa,
                                                      const float* __restrict__ xb, int sub, float& res_a, float& res_b,
                                                      float sa = 1.0f, float sb = 1.0f) {
  const float4* a4 = (const float4*)xa + sub;
  const float4* b4 = (const float4*)xb + sub;
  const float4* q4 = (const float4*)qp + sub;
  float4 ra[N16], rb[N16];
#pragma unroll
  for (int i = 0; i < N16; ++i) ra[i] = a4[i * 4];
#pragma unroll
  for (int i = 0; i < N16; ++i) rb[i] = b4[i * 4];
  float pa = 0.0f, pb = 0.0f;
#pragma unroll
  for (int i = 0; i < N16; ++i) canon_group_step<METRIC01, SCALE>(ra[i], q4[i * 4], pa, 4, sa);
#pragma unroll
  for (int i = 0; i < N16; ++i) canon_group_step<METRIC01, SCALE>(rb[i], q4[i * 4], pb, 4, sb);
  const float a0 = __shfl(pa, 0, 4), a1 = __shfl(pa, 1, 4), a2 = __shfl(pa, 2, 4), a3 = __shfl(pa, 3, 4);
  const float b0 = __shfl(pb, 0, 4), b1 = __shfl(pb, 1, 4), b2 = __shfl(pb, 2, 4), b3 = __shfl(pb, 3, 4);
  res_a = ex_add(ex_add(ex_add(a0, a1), a2), a3);
  res_b = ex_add(ex_add(ex_add(b0, b1), b2), b3);
  if (METRIC01 != 0) {
    res_a = ex_sub(1.0f, res_a);
    res_b = ex_sub(1.0f, res_b);
  }
}

// Four rows of exactly 16 * N16 floats (N16 <= 8: rows of up to 128 dims) by one 4-lane group, all 4 * N16 loads of the lane
// in flight before the first product: 64 rows per pass of a wave (the wide graph walk, k_graphw.hip, evaluates up to 64
// fresh rows per merge — one memory round trip instead of two).  Same arithmetic and order per row as canon_dist_group_t.
template <int METRIC01, int N16, bool SCALE = false>
__device__ __forceinline__ void canon_dist_group_quad(const float* __restrict__ qp, const float* const (&x)[4], int sub,
                                                      float (&res)[4], const float (&sc)[4]) {
  const float4* q4 = (const float4*)qp + sub;
  float4 r[4][N16];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float4* x4 = (const float4*)x[j] + sub;
#pragma unroll
    for (int i = 0; i < N16; ++i) r[j][i] = x4[i * 4];
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float p = 0.0f;
#pragma unroll
    for (int i = 0; i < N16; ++i) canon_group_step<METRIC01, SCALE>(r[j][i], q4[i * 4], p, 4, sc[j]);
    const float t0 = __shfl(p, 0, 4), t1 = __shfl(p, 1, 4), t2 = __shfl(p, 2, 4), t3 = __shfl(p, 3, 4);
    res[j] = ex_add(ex_add(ex_add(t0, t1), t2), t3);
    if (METRIC01 != 0) res[j] = ex_sub(1.0f, res[j]);
  }
}

// Canonical distances of the search-copy rows ids_l[0..count) (LDS) to the permuted query qs (LDS), by one wave:
// lane p (< count) returns the distance of row p, other lanes +inf.  16 rows per pass, one 4-lane group per row
// (canon_dist_group_t); rows of 32 / 64 / 96 / 128 / 192 / 256 dims go 32 per pass, two per group, with the loads
// of both rows in flight together (canon_dist_group_pair) — at 128 dims and 27 fresh neighbours per expansion that is
// one memory round trip per expansion instead of two (6.25 M x 128-class workloads: -20 % kernel time).
// xscale (optional): per-row scale applied to the row's elements on the fly (single-copy graph spaces, cosine:
// inv_norm) — nullptr: the rows are used as stored.
// QUAD (the wide graph walk): more than 32 rows of 32 / 64 / 96 / 128 dims go 64 per pass, four per group.
template <int METRIC01, bool SCALE, bool QUAD = false>
__device__ __forceinline__ float wave_group_dists_t(const float* __restrict__ qs, const float* __restrict__ Xs, uint32_t ld,
                                                    uint32_t dims, const uint32_t* ids_l, uint32_t count, int lane,
                                                    const float* __restrict__ xscale) {
  float mine = __builtin_inff();
  const bool pairable = dims <= 256 && (dims == 32 || dims == 64 || dims == 96 || dims == 128 || dims == 192 || dims == 256);
  if (QUAD && pairable && dims <= 128 && count > 32) {   // (count <= 64: one pass)
    const uint32_t r0 = (uint32_t)lane >> 2;
    const float* x[4];
    float sc[4], res[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t rj = r0 + 16u * (uint32_t)j;
      const uint32_t id = ids_l[rj < count ? rj : r0];   // a missing row: the group's first one again, result dropped
      sc[j] = SCALE ? __hip_atomic_load(xscale + id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) : 1.0f;
      x[j] = Xs + (size_t)id * ld;
    }
    const int sub = lane & 3;
    switch (dims) {
      case 32: canon_dist_group_quad<METRIC01, 2, SCALE>(qs, x, sub, res, sc); break;
      case 64: canon_dist_group_quad<METRIC01, 4, SCALE>(qs, x, sub, res, sc); break;
      case 96: canon_dist_group_quad<METRIC01, 6, SCALE>(qs, x, sub, res, sc); break;
      default: canon_dist_group_quad<METRIC01, 8, SCALE>(qs, x, sub, res, sc); break;
    }
    const int src = (lane & 15) << 2;
    const float g0 = __shfl(res[0], src, 64), g1 = __shfl(res[1], src, 64), g2 = __shfl(res[2], src, 64),
                g3 = __shfl(res[3], src, 64);
    if ((uint32_t)lane < count) mine = (lane & 32) ? ((lane & 16) ? g3 : g2) : ((lane & 16) ? g1 : g0);
    return mine;
  }
  if (pairable && count > 16) {
    for (uint32_t base = 0; base < count; base += 32) {
      const uint32_t ra = base + ((uint32_t)lane >> 2), rb = ra + 16;
      float res_a = __builtin_inff(), res_b = __builtin_inff();
      if (ra < count) {
        const bool have_b = rb < count;  // a missing second row: the first one again, result dropped
        const uint32_t ia = ids_l[ra], ib = ids_l[have_b ? rb : ra];
        const float sa = SCALE ? __hip_atomic_load(xscale + ia, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) : 1.0f;
        const float sb = SCALE ? __hip_atomic_load(xscale + ib, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) : 1.0f;
        const float* xa = Xs + (size_t)ia * ld;
        const float* xb = Xs + (size_t)ib * ld;
        const int sub = lane & 3;
        switch (dims) {
          case 32: canon_dist_group_pair<METRIC01, 2, SCALE>(qs, xa, xb, sub, res_a, res_b, sa, sb); break;
          case 64: canon_dist_group_pair<METRIC01, 4, SCALE>(qs, xa, xb, sub, res_a, res_b, sa, sb); break;
          case 96: canon_dist_group_pair<METRIC01, 6, SCALE>(qs, xa, xb, sub, res_a, res_b, sa, sb); break;
          case 128: canon_dist_group_pair<METRIC01, 8, SCALE>(qs, xa, xb, sub, res_a, res_b, sa, sb); break;
          case 192: canon_dist_group_pair<METRIC01, 12, SCALE>(qs, xa, xb, sub, res_a, res_b, sa, sb); break;
          default: canon_dist_group_pair<METRIC01, 16, SCALE>(qs, xa, xb, sub, res_a, res_b, sa, sb); break;
        }
        if (!have_b) res_b = __builtin_inff();
      }
      const float got_a = __shfl(res_a, (lane & 15) << 2, 64), got_b = __shfl(res_b, (lane & 15) << 2, 64);
      if (((uint32_t)lane & ~31u) == base && (uint32_t)lane < count) mine = (lane & 16) ? got_b : got_a;
    }
    return mine;
  }
  for (uint32_t base = 0; base < count; base += 16) {
    const uint32_t r = base + ((uint32_t)lane >> 2);
    float res = __builtin_inff();
    if (r < count) {
      const uint32_t id = ids_l[r];
      // (the scale is requested FIRST, as an ordered load: it is the oldest entry of the load queue when the first
      // product needs it — sunk below the row's ring loads it would be the youngest, and waiting for it would drain
      // the ring)
      const float xsc = SCALE ? __hip_atomic_load(xscale + id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) : 1.0f;
      res = canon_dist_group_t<METRIC01, SCALE>(qs, Xs + (size_t)id * ld, lane & 3, dims, xsc);
    }
    const float got = __shfl(res, (lane & 15) << 2, 64);
    if (((uint32_t)lane & ~15u) == base && (uint32_t)lane < count) mine = got;
  }
  return mine;
}
template <int METRIC01, bool QUAD = false>
__device__ __forceinline__ float wave_group_dists(const float* __restrict__ qs, const float* __restrict__ Xs, uint32_t ld,
                                                  uint32_t dims, const uint32_t* ids_l, uint32_t count, int lane,
                                                  const float* __restrict__ xscale = nullptr) {
  if (METRIC01 == 1 && xscale) return wave_group_dists_t<METRIC01, true, QUAD>(qs, Xs, ld, dims, ids_l, count, lane, xscale);
  return wave_group_dists_t<METRIC01, false, QUAD>(qs, Xs, ld, dims, ids_l, count, lane, nullptr);
}

// runtime-dispatch form (metric: 0 = L2^2, 1 = 1 - inner product; scale_x: cosine rows) used by the
// insertion kernels (EHX_INSERT_RING: with or without the register ring — A/B switch)
#ifndef EHX_INSERT_RING
#define EHX_INSERT_RING 0
#endif
__device__ __forceinline__ float canon_dist_lane(int metric, const float* __restrict__ q,
                                                 const float* __restrict__ x, float xscale, bool scale_x,
                                                 uint32_t dims) {
  if (metric == 0) return canon_dist_lane_t<0, false, EHX_INSERT_RING != 0>(q, x, xscale, dims);
  if (scale_x) return canon_dist_lane_t<1, true, EHX_INSERT_RING != 0>(q, x, xscale, dims);
  return canon_dist_lane_t<1, false, EHX_INSERT_RING != 0>(q, x, xscale, dims);
}

struct ScanArgs {
  const float* Q;        // [q_tiles*256][ld] prepared queries (zero padded)
  const void* X;         // [cap][ld] stored rows (fp32, or fp16 when x_half), cap % 256 == 0, pad columns zero
  uint32_t x_half;       // 1: rows are IEEE fp16
  const float2* rowp;    // [cap] epilogue (a, b): approx distance = dot*a + b
  uint64_t* cand;        // [grid][256][64] per-block candidate slots (scratch)
  uint64_t* part;        // [q_tiles*256][n_chunks][lists_per_chunk][kprime] sorted partial top-k' keys
  uint32_t n;            // valid rows
  uint32_t ld;           // row stride in floats, % 32 == 0
  uint32_t tile0;        // first tile of this pass
  uint32_t n_tiles;      // tiles of this pass (tile = 128 rows)
  uint32_t list0;        // first sorted-list slot of this pass in `part`
  uint32_t lists_total;  // sorted lists per query in `part` (all passes)
  uint32_t q_tiles;
  uint32_t n_chunks;
  uint32_t tiles_per_chunk;
  uint32_t kprime;       // <= 64
  uint32_t* err;         // device error counter (bounded-retry guard tripped)
  unsigned long long* gthr;  // [q_tiles*256] global per-query threshold keys (init ~0 per launch; 8-wave kernel)
  uint32_t xcd_map;      // 1: blocks of one chunk share an XCD (grid % 8 == 0, n_chunks % 8 == 0)
};

uint32_t scan_lists_per_chunk();  // sorted key lists each (query, chunk) publishes: 2 (one per wave row)
hipError_t launch_flat_scan(const ScanArgs& a, hipStream_t st);   // the fp32 matrix-core scan
hipError_t launch_flat_scan8(const ScanArgs& a, hipStream_t st);  // k_flat8.hip: 8 waves, two per SIMD

// ---- fp16-MFMA filter scan (k_flat16.hip) ----
struct ScanArgs16 {
  const __half* Q;       // [q_tiles*256][ld] unit-normalised queries, binary16 (zero padded), scan16_index layout
  const __half* X;       // [cap][ld] scan copy: unit-normalised rows, binary16, scan16_index layout; cap % 256 == 0
  const float2* rowp;    // [cap] (a_r, b_r): S = b_r*gamma_q + a_r*dot;  padding rows (0, +inf)
  const float* qgamma;   // [q_tiles*256] gamma_q
  float eps;             // accumulator start value: bound of |dot16 - true dot|
  uint32_t cos;          // 1: every valid row has (a, b) = (-1, 1) and gamma = 1 (cosine): fast phase 1
  float* dump = nullptr; // sample pass: write every score to dump[row - tile0*256][q_tiles*256] instead of keeping lists
  uint64_t* cand;
  uint64_t* part;
  uint32_t n;
  uint32_t ld;           // row stride in halves, % 128 == 0 (a tile = a whole number of LDS ring revolutions)
  uint32_t tile0, n_tiles, list0, lists_total, q_tiles, n_chunks, tiles_per_chunk, kprime;
  uint32_t* err;
  unsigned long long* gthr;
  uint32_t xcd_map;
};
// Stage-blocked layout of the fp16 scan copy and of the fp16 query tiles: the matrix is cut into tiles of
// 256 rows and stages of 32 columns; one (tile, stage) block is 256 rows x 64 bytes = 16 KiB, stored
// contiguously in exactly the image the kernel wants in LDS (16-byte chunk c of row r at physical chunk
// c ^ ((r>>2)&3)), blocks ordered [tile][stage].  A stage's DMA is then a linear 16-KiB copy: every
// global->LDS instruction moves 1 KiB = 8 full cache lines.  Index (in halves) of element (row, col):
__host__ __device__ inline size_t scan16_index(uint64_t row, uint32_t col, uint32_t ld16) {
  const uint64_t tile = row >> 8;
  const uint32_t rr = (uint32_t)(row & 255u), kt = col >> 5, cc = col & 31u;
  const uint32_t chunk = (cc >> 3) ^ ((rr >> 2) & 3u);
  return ((size_t)(tile * (ld16 >> 5) + kt) * 256u + rr) * 32u + chunk * 8u + (cc & 7u);
}
// The fp16 query tiles use the same blocks, [q_tile][stage], with stages 0..2 of every tile stored once more
// after its last stage (the kernel's DMA runs three stages ahead and wraps into the next row tile without
// re-basing its query pointer mid-tile): (ld16/32 + 3) blocks per query tile.
__host__ __device__ inline size_t scanq16_index(uint64_t row, uint32_t stage, uint32_t cc, uint32_t ld16) {
  const uint64_t tile = row >> 8;
  const uint32_t rr = (uint32_t)(row & 255u);
  const uint32_t chunk = (cc >> 3) ^ ((rr >> 2) & 3u);
  return ((size_t)(tile * ((ld16 >> 5) + 3u) + stage) * 256u + rr) * 32u + chunk * 8u + (cc & 7u);
}
inline size_t scanq16_halves(uint32_t q_rows, uint32_t ld16) { return (size_t)(q_rows >> 8) * ((ld16 >> 5) + 3u) * 256u * 32u; }
constexpr size_t kScan16TailPadHalves = 3u * 256u * 32u;  // X16 tail padding: three stage blocks (DMA read-ahead)
size_t scan16_lds_bytes();
hipError_t launch_flat_scan16(const ScanArgs16& a, hipStream_t st);
// bound of |<fp16(q^), fp16(x^)> accumulated in fp32 - <q^, x^>| for unit vectors of `dims` elements
inline float scan16_eps(uint32_t dims) { return 1.0e-3f + 2.0e-7f * (float)dims; }

// ---- int8-MFMA filter scan (k_flati8.hip) ----
constexpr uint32_t kPoolCap = 4096;    // candidate keys one query can collect in one pass (overflow: query flagged)
constexpr uint32_t kSyncWordsI8 = 1024; // lock-step progress words of the int8 scan: [n_chunks <= 256][4 query tiles]
constexpr uint32_t kMerged8 = 256;     // default width of the running best list of the int8 pipeline
constexpr uint32_t kMerged8Max = 1024; // widest list (a space widens its list when queries go uncertified: ehx_flat.cpp)
struct ScanArgsI8 {
  const int8_t* Q;        // [q_tiles][ld/64 + 3][256][64] int8 query tiles, scan8 stage-blocked layout
  const int8_t* X;        // scan copy, scan8_index layout; cap % 256 == 0
  const float4* rowp;     // [cap + 512] (A, B, C, D) per row; padding rows (0, +inf, 0, 0)
  const float4* tilep;    // [cap/256 + 2] (max|A|, max|C|, max|D|, min B) per 256-row tile
  const float* tileg;     // [cap/256 + 2][16]: [0..8) max |A| of each 32-row lane group of the tile (g = 4 wr + (l >> 4)),
                          // [8..16) the group's B margin: min B of the group - min B of the tile (>= 0; round 6)
  const uint8_t* perm;    // [cap] row index inside its tile of the row stored at each position (identity: unsorted tile)
  const float4* qparams;  // [q_tiles*256] (s_q, e_q, gamma_q, smallest threshold the query was scanned with so far)
  const float* thr;       // [q_tiles*256] score threshold of this pass per query (-inf: padding query)
  float* dump = nullptr;  // sample pass: every lower bound -> dump[scan8_dump_index(q, row - tile0*256, rows of the sample)]
  uint64_t* cand;         // [grid][512][64] staging slots
  uint64_t* pool;         // [q_tiles*256][pool_cap] collected (score, id) keys of this pass, unsorted
  uint32_t* pool_cnt;     // [q_tiles*256]
  uint32_t* ovf;          // [q_tiles*256] 1 = the pool overflowed: the query must be answered by another engine
  uint32_t pool_cap;
  uint32_t n;
  uint32_t ld;            // bytes (= k-values) per row of the scan copy, % 64 == 0 (whole 64-byte stages)
  uint32_t tile0, n_tiles, q_tiles, n_chunks, tiles_per_chunk;
  uint32_t xcd_map;
  uint32_t* sync = nullptr;  // lock-step words, zero before the launch (nullptr: off): sync_tol == 0: [n_chunks] counters,
                             // one add per workgroup and ring revolution; sync_tol > 0: [n_chunks][4] progress words,
                             // tiles completed by each of the chunk's (<= 4) query-tile workgroups
  uint32_t sync_tol = 0;     // > 0: a workgroup does not run more than this many TILES ahead of its slowest sibling
  uint32_t skew = 0;         // half-tile workgroups: the second-resident wave of a SIMD starts skew x 64 cycles late
  uint32_t group_b = 0;      // 1: the alarm level of a lane group uses the group's B margin (L2^2: B_r = |x_r|^2 varies)
};
// Stage-blocked layout of the int8 scan copy / query tiles: tiles of 256 rows, stages of 64 columns (bytes); one
// (tile, stage) block is 256 rows x 64 B = 16 KiB in exactly the LDS image of the kernel (16-byte chunk c of row r
// at physical chunk c ^ ((r>>2)&3)); blocks ordered [tile][stage].  Byte index of element (row, col):
// Chunk swizzle of a row: the scan reads a block's fragment with ONE ds_read_b128 — lane l takes chunk l >> 4 of row
// l & 15 — and the LDS serves that instruction in four groups of 16 lanes ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and
// the same + 32: MI355X_MICROARCH.md, LDS).  With g = (0, 2, 3, 1)[(row >> 2) & 3] the 16 lanes of every group fall
// into 16 different 16-byte slots of the 256-byte LDS row (rows r, r+4, r+8, r+12 of a group carry chunks that differ
// after the XOR): conflict-free.  (Rounds 2-3 used (row >> 2) & 3, made for the 32x32x32 fragment's lane pattern.)
__host__ __device__ inline uint32_t scan8_swz(uint32_t rr) { return (0x78u >> (((rr >> 2) & 3u) * 2u)) & 3u; }
__host__ __device__ inline size_t scan8_index(uint64_t row, uint32_t col, uint32_t ld8) {
  const uint64_t tile = row >> 8;
  const uint32_t rr = (uint32_t)(row & 255u), kt = col >> 6, cc = col & 63u;
  const uint32_t chunk = (cc >> 4) ^ scan8_swz(rr);
  return ((size_t)(tile * (ld8 >> 6) + kt) * 256u + rr) * 64u + chunk * 16u + (cc & 15u);
}
// query tiles: the same blocks, [q_tile][stage], stages 0..2 stored once more after the last (DMA read-ahead)
__host__ __device__ inline size_t scanq8_index(uint64_t row, uint32_t stage, uint32_t cc, uint32_t ld8) {
  const uint64_t tile = row >> 8;
  const uint32_t rr = (uint32_t)(row & 255u);
  const uint32_t chunk = (cc >> 4) ^ scan8_swz(rr);
  return ((size_t)(tile * ((ld8 >> 6) + 3u) + stage) * 256u + rr) * 64u + chunk * 16u + (cc & 15u);
}
// the sample pass's dump of lower bounds (flat_scan_i8_kernel<DUMP> -> sample_select256_kernel): blocks of 16 queries x 16
// sample rows = 1 KiB, inside a block a query's 16 rows are contiguous; element index of (query q, sample row `row`) when
// the sample holds n_s rows (n_s % 16 == 0)
__host__ __device__ inline size_t scan8_dump_index(uint32_t q, uint32_t row, uint32_t n_s) {
  return ((((size_t)(q >> 4) * (n_s >> 4)) + (row >> 4)) << 8) + ((q & 15u) << 4) + (row & 15u);
}
inline size_t scanq8_bytes(uint32_t q_rows, uint32_t ld8) { return (size_t)(q_rows >> 8) * ((ld8 >> 6) + 3u) * 256u * 64u; }
constexpr size_t kScan8TailPadBytes = 3u * 256u * 64u;  // X8 tail padding: three stage blocks (DMA read-ahead)
size_t scan_i8_lds_bytes();
hipError_t launch_flat_scan_i8(const ScanArgsI8& a, hipStream_t st);
// scan copy of rows [row0, row0+n): X8 = int8(x/|x| / s_r), rowp8 = (A, B, C, D); then the tile parameters of
// every tile touching the range.  n_unsafe[2]: [0] += rows the filter cannot bound, [1] += lane groups whose min B lies more
// than 0.1 % above their tile's (the scan uses the groups' B margins only in spaces that have any).
// Full tiles inside [sort_lo, sort_hi) (and inside the rows written) are stored ordered by quantisation step, every row's
// step raised to its 32-row lane group's maximum (perm8[position] = row index inside the tile, tileg8[tile][8 of 16] =
// |A| of each group: k_misc.hip, "rows of a tile ordered by quantisation step"); every other touched tile keeps the
// row order and the rows' own steps.  The caller guarantees that no scan can read the tiles of [sort_lo, sort_hi)
// meanwhile.  scratch: device memory of make_scan8_scratch_bytes(...) bytes.
size_t make_scan8_scratch_bytes(uint64_t row0, uint64_t n, uint64_t sort_lo, uint64_t sort_hi);
hipError_t launch_make_scan8(const void* X, int x_half, uint64_t row0, uint64_t n, uint32_t dims, uint32_t ld,
                             uint32_t ld8, int metric, int8_t* X8, float4* rowp8, float4* tilep8, uint8_t* perm8,
                             float* tileg8, uint64_t sort_lo, uint64_t sort_hi, void* scratch,
                             unsigned long long* n_unsafe, hipStream_t st);
hipError_t launch_tile_ids(uint64_t* out, uint64_t t0, uint64_t t1, uint64_t a0, uint64_t a1, hipStream_t st);
// perm8 of rows [row0, row0+n): identity
hipError_t launch_perm8_pad(uint8_t* perm8, uint64_t row0, uint64_t n, hipStream_t st);
// rowp8 for padding rows [row0, row0+n): (0, +inf, 0, 0); tilep8 for padding tiles [t0, t0+n): never alarm
hipError_t launch_rowp8_pad(float4* rowp8, uint64_t row0, uint64_t n, hipStream_t st);
hipError_t launch_tilep8_pad(float4* tilep8, uint64_t t0, uint64_t n, hipStream_t st);
// int8 query tiles + (s_q, e_q, gamma_q) + (u, v) with D = u*S + v; thr[q] = +inf for q < nq, -inf for padding
// ... in ONE launch with the prepared fp32 rows of the re-rank (launch_prep_queries' q_out) and the zeroing of the scan's
// control words ctl = [q_rows] pool counts | [q_rows] overflow flags | [256] lock-step counters
hipError_t launch_prep_queries_i8(const float* q_in, uint32_t nq, uint32_t dims, uint32_t ld, uint32_t ld8,
                                  uint32_t q_rows, int metric, float* q_out, int8_t* Q8, float4* qparams, float2* quv,
                                  float* thr, uint32_t* ctl, hipStream_t st);
// sample pass -> first thresholds: thr[q] = the rank-th (<= 64) smallest of scores[0..n_rows)[q] (+inf if fewer)
hipError_t launch_sample_select256(const float* scores, uint32_t n_rows, uint32_t q_rows, uint32_t nq,
                                   uint32_t rank, float* thr, hipStream_t st);
// merge one pass's pool into the query's running best kMerged8 keys (seed: keep what `merged` holds); publishes
// thr[q] = score of the kprime-th best (+inf while fewer are known), lowers qparams[q].w to the threshold the merged
// pass was scanned with, and empties the pool (pool_cnt = 0)
// (width: the list's length, a power of two in [256, kMerged8Max], kprime <= width)
hipError_t launch_select256(const uint64_t* pool, uint32_t* pool_cnt, uint32_t pool_cap, uint32_t nq, uint32_t kprime,
                            uint64_t* merged, uint32_t width, bool seed, float* thr, float4* qparams, hipStream_t st);
struct Rerank256Args {
  const float* Q;          // prepared (canonical) queries [*][ld]
  const void* X;
  uint32_t x_half;
  const float* inv_norm;
  const uint64_t* merged;  // [nq][width] keys (S_lower, id), ascending
  uint32_t width;          // list stride
  const uint32_t* ovf;     // [nq] pool overflow flags
  const float2* quv;       // [nq] D = u*S + v
  const float4* qparams;   // [nq] .w = the smallest threshold the query was scanned with (select256_kernel)
  const float* max_sumsq;
  uint64_t* out_ids;
  float* out_dist;
  uint32_t* out_count;
  unsigned long long* n_uncertified;
  uint32_t* uncert_flags;
  uint32_t nq, k, kprime, n, dims, ld;
  int metric;
};
hipError_t launch_rerank256(const Rerank256Args& a, hipStream_t st);

// sample pass -> starting thresholds: gthr[q] = key of the kprime-th smallest of scores[0..n_rows)[q] (id part
// 0xFFFFFFFF, so a row that ties the threshold still passes); one wave per query
hipError_t launch_sample_select(const float* scores, uint32_t n_rows, uint32_t q_rows, uint32_t nq, uint32_t kprime,
                                unsigned long long* gthr, hipStream_t st);

// scan copy of rows [row0, row0+n): X16 = fp16(x/|x|), rowp16 = (a_r, b_r) per metric.  Rows the filter
// cannot bound (non-finite or denormal-range norms) are counted in *n_unsafe (the space then stays on
// the fp32 scan).
hipError_t launch_make_scan16(const void* X, int x_half, uint64_t row0, uint64_t n, uint32_t dims, uint32_t ld,
                              uint32_t ld16, int metric, __half* X16, float2* rowp16, unsigned long long* n_unsafe,
                              hipStream_t st);
// filter-side query preparation: Q16 = fp16(q/|q|) padded to [q_rows][ld16]; qgamma[q]; quv[q] = (u, v) with
// D = u*S + v.  Queries the filter cannot bound get u = NaN (never certified -> fp32 re-run).
hipError_t launch_prep_queries16(const float* q_in, uint32_t nq, uint32_t dims, uint32_t ld16, uint32_t q_rows,
                                 int metric, __half* Q16, float* qgamma, float2* quv, hipStream_t st);

hipError_t launch_set_gthr(const uint64_t* merged, uint32_t nq, uint32_t kprime, unsigned long long* gthr, hipStream_t st);

// one wave per query: k-way merge of the per-chunk sorted key lists -> top-kprime keys
// (merges `n_chunks` consecutive lists of each query; a query's lists are `lists_stride` apart)
// seed: start from the keys already in `merged` (earlier passes); gthr (optional): publish the k'-th best
hipError_t launch_flat_merge(const uint64_t* part, uint32_t nq, uint32_t n_chunks, uint32_t kprime,
                             uint64_t* merged /*[nq][64]*/, hipStream_t st, uint32_t lists_stride, bool seed = false,
                             unsigned long long* gthr = nullptr);

// canonical (oracle-order) distances of the merged candidates, sort by (dist, id), emit top-k.
struct RerankArgs {
  const float* Q;          // prepared queries [*][ld]
  const void* X;           // fp32 or fp16 rows
  uint32_t x_half;
  const float* inv_norm;   // [cap] (cosine) or nullptr
  const uint64_t* merged;  // [nq][64] keys (approx score, id)
  uint64_t* out_ids;       // [nq][k]
  float* out_dist;         // [nq][k]
  uint32_t* out_count;     // [nq]
  unsigned long long* n_uncertified;  // device counter
  uint32_t nq, k, kprime, n, dims, ld;
  int metric;
  // fp16-filter scans: the keys hold S_lower; D = u*S + v maps the worst candidate back to a distance.
  // nullptr for the fp32 scan.
  const float2* quv = nullptr;
  uint32_t* uncert_flags = nullptr;  // [nq] 1 = not certified (optional)
  uint32_t exact_keys = 0;           // keys come from launch_exhaustive (exact distances): skip the certification
  const float* max_sumsq = nullptr;  // largest |x|^2 in the space (device scalar, launch_row_stats): margin of the certificate
  uint32_t out_stride = 0, out_offset = 0;  // paged output: row stride (0 = k) and first column of this page
};
hipError_t launch_rerank(const RerankArgs& a, hipStream_t st);
// canonical distance of every row for each of nq prepared queries: out[q][block][64] best (distance, id) keys
// (floor, optional: per query, only keys strictly above floor[q] are kept — paging for k > 64)
hipError_t launch_exhaustive(const float* Q, const void* X, int x_half, const float* inv_norm, uint32_t n, uint32_t dims,
                             uint32_t ld, int metric, uint32_t rows_per_block, uint32_t n_blocks, uint32_t nq,
                             const uint64_t* floor, uint64_t* out, hipStream_t st);
hipError_t launch_set_floor(const uint64_t* merged, uint32_t nq, uint64_t* floor, hipStream_t st);
// one query from host-visible memory against a small shard in one launch (k_flat.hip: single_query_kernel)
struct SingleQueryArgs {
  const float* q_in;        // [dims] raw query (host-visible pinned memory, or device memory)
  const void* X;            // rows, fp32 or fp16 (x_half)
  const float* inv_norm;    // [cap] (cosine)
  uint64_t* part;           // [n_blocks][64] scratch: every workgroup's best keys
  uint32_t* ticket;         // device counter, 0 between calls
  uint64_t* out_ids;        // [k]  host-visible
  float* out_dist;          // [k]  host-visible
  uint32_t* out_count;      // [1]  host-visible
  uint32_t* done_flag;      // host-visible: set to `seq` when the results are in place
  uint32_t seq, x_half, n, dims, ld, rows_per_block, k;
  int metric;
};
hipError_t launch_single_query(const SingleQueryArgs& a, uint32_t n_blocks, hipStream_t st);

// prepared queries: copy into the padded [q_rows][ld] buffer, L2-normalise for cosine
hipError_t launch_prep_queries(const float* q_in, uint32_t nq, uint32_t dims, uint32_t ld,
                               uint32_t q_rows, int metric, float* q_out, hipStream_t st);

// sub-batches of the engine chain: rows idx[0..m) of the caller's query batch gathered into a dense [m][dims] matrix,
// and a sub-batch's results written back to the rows they belong to
hipError_t launch_gather_queries(const float* src, const uint32_t* idx, uint32_t m, uint32_t dims, float* dst,
                                 hipStream_t st);
hipError_t launch_scatter_results(const uint64_t* ids, const float* dist, const uint32_t* cnt, const uint32_t* idx,
                                  uint32_t m, uint32_t k, uint64_t* out_ids, float* out_dist, uint32_t* out_cnt,
                                  hipStream_t st);

// per-row statistics for rows [row0, row0+n): inv_norm (cosine), rowp (a,b) for the scan epilogue;
// *max_sumsq (optional) is raised to the largest |x|^2 seen (the certification margin's norm bound)
// perm: the fp32 rows are stored block-permuted (single-copy graph spaces); the sums keep the logical order
hipError_t launch_row_stats(const void* X, int x_half, uint64_t row0, uint64_t n, uint32_t dims, uint32_t ld,
                            int metric, float* inv_norm, float2* rowp, float* max_sumsq, hipStream_t st, int perm = 0);
// single-copy graph spaces: rows [row0, row0 + n) (or rows ids[0..n)) into the search copy's block order, in place
hipError_t launch_permute_blocks(float* X, uint32_t ld, uint64_t row0, uint64_t n, const uint64_t* ids, hipStream_t st);
// rowp for padding rows [row0, row0+n): (0, +inf)
hipError_t launch_rowp_pad(float2* rowp, uint64_t row0, uint64_t n, hipStream_t st);
// graph mode: rows [row0, row0+n) of the search copy (16-float blocks permuted for the four SSE partial
// sums, cosine rows pre-normalised); must follow launch_row_stats (inv_norm)
hipError_t launch_make_search_copy(const void* X, bool x_half, const float* inv_norm, uint64_t row0, uint64_t n,
                                   uint32_t ld, int metric, float* Xs, hipStream_t st);

// fp16 storage: rows of an fp32 matrix (stride src_ld) rounded to nearest-even into rows ids[i] (or
// row0+i when ids == nullptr) of the fp16 matrix, and one fp16 row widened back for Get
hipError_t launch_store_rows_f16(const float* src, uint32_t src_ld, const uint64_t* ids, uint64_t row0, uint64_t n,
                                 uint32_t dims, uint32_t ld, __half* X, hipStream_t st);
hipError_t launch_load_row_f16(const __half* X, uint64_t row, uint32_t dims, uint32_t ld, float* out, hipStream_t st);

// EHX-GAUSS-1 rows row0, row0 + row_stride, ... generated straight into a [*, ld] matrix (optionally L2-normalised)
// latent != 0: EHX-MANIFOLD-1 rows on a `latent`-dimensional linear subspace + 5 % noise (include/ehx_datagen.h)
hipError_t launch_gen_rows(uint64_t seed, uint64_t row0, uint64_t n_rows, uint32_t dims, uint32_t ld,
                           int normalize, float* out, hipStream_t st, uint64_t row_stride = 1, uint32_t latent = 0);

// graph-mode search (k_graph.hip): one wave per query
constexpr uint32_t kGraphCounters = 12;  // n_dist, n_hops0, n_hops_up, n_prefetch_hit, [4..11] profile builds (wide walk: [4] = steps)
struct GraphArgs {
  const float* Q;           // prepared queries [nq][ld]
  const float* X;           // rows [cap][ld]
  const float* Xs;          // search copy [cap][ld] (launch_make_search_copy)
  const float* inv_norm;    // cosine
  const float* xscale = nullptr;  // single-copy graph spaces, cosine: Xs holds RAW permuted rows, scaled by inv_norm on the fly
  const uint32_t* adj0;     // [n][M0], pad 0xFFFFFFFF, stored order
  const uint32_t* up_start; // [n]: first upper list of the node (levels 1..L consecutive) or ~0
  const uint32_t* up_lists; // [*][M], pad 0xFFFFFFFF
  uint32_t* visited;        // [nq][vis_words], all-zero before the launch and again after it
  uint32_t* vislog;         // [nq][vislog_cap] rows a query marked (it clears their words when done)
  uint32_t vislog_cap;
  uint64_t* out_ids;        // [nq][k]
  float* out_dist;
  uint32_t* out_count;
  unsigned long long* counters;  // [kGraphCounters]
  uint32_t nq, k, ef, ef_cap, n, dims, ld, M, M0, vis_words, entry_point;
  int max_level, metric;
  // One query per call in ONE launch (round 5; the reference's request shape, server.cc:172-210): q_raw != nullptr — the
  // raw query is read from host-visible memory and prepared by the kernel itself into Q (device scratch, [ld]); out_*
  // then point into host-visible memory and the kernel publishes `seq` in done_flag (system scope) when they are written.
  const float* q_raw = nullptr;
  uint32_t* done_flag = nullptr;
  uint32_t seq = 0;
  // Expansions per step of the level-0 search (ehx_params.search_width): 1 = the strict walk (hnswlib's order, one node at
  // a time: graph_search_kernel); 2 / 4 = the wide walk (k_graphw.hip: the `width` closest unexpanded entries of the
  // result list are expanded together — same ef bound and termination, fewer dependent memory round trips per query).
  uint32_t width = 1;
};
size_t graph_lds_bytes(uint32_t ld, uint32_t ef_cap, uint32_t width = 1);
hipError_t launch_graph_search(const GraphArgs& a, hipStream_t st);

// graph-mode insertion (k_insert.hip)
// hipFuncAttributeMaxDynamicSharedMemorySize is set per function AND per device (every device has its own loaded code
// object), and launches come from several host threads once a space is sharded over devices inside one process: the
// largest size set so far is kept per device, atomically.  `fns`: the kernel's instantiations.
struct DynLdsAttr {
  std::atomic<size_t> set[64] = {};
  hipError_t ensure(const void* const* fns, int n_fns, size_t bytes) {
    if (bytes <= 64 * 1024) return hipSuccess;  // (within the default limit)
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::atomic<size_t>& cur = set[dev & 63];
    if (cur.load(std::memory_order_acquire) >= bytes) return hipSuccess;
    for (int i = 0; i < n_fns; ++i) {
      e = hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
      if (e != hipSuccess) return e;
    }
    size_t seen = cur.load(std::memory_order_relaxed);
    while (seen < bytes && !cur.compare_exchange_weak(seen, bytes, std::memory_order_release)) {
    }
    return hipSuccess;
  }
};

struct InsertArgs {
  const float* X;
  const float* Xs;         // search copy (launch_make_search_copy)
  const float* inv_norm;
  const float* xscale;     // single-copy graph spaces, cosine: Xs holds RAW permuted rows, scaled by inv_norm on the fly (else nullptr)
  uint32_t* adj0;          // [cap][M0]
  uint32_t* up_start;      // [cap]
  uint32_t* up_lists;      // [*][M]
  uint32_t* visited;       // [P][vis_words], zero on entry (and on exit)
  uint32_t* vislog;        // [P][vislog_cap]
  const uint32_t* new_ids; // [P]
  const int32_t* new_levels;
  uint32_t* sel;           // [P][max_sel_levels][1+M]: per level (count, ids farthest first)
  uint32_t ef, dims, ld, M, M0, vis_words, vislog_cap, max_sel_levels, entry_point;
  int max_level, metric;
  uint32_t exclude_self;   // 1: repairConnectionsForUpdate (drop the node itself from the search results)
  // ---- bulk build: the link work items are made on the device (no host between the search and the link kernel) ----
  uint32_t id0;            // new_ids == nullptr: wave p inserts row id0 + p
  uint32_t head_rows;      // list id of (node t, level l): l == 0 ? t : head_rows + up_start[t] + l - 1
  uint32_t* link_head;     // [head_rows + upper lists], 0 = no incoming link this round; otherwise 1 + the pair
                           // (wave p, level, slot) = (p * max_sel_levels + level) * M + slot registered last on the list
  uint32_t* link_next;     // [P * max_sel_levels * M]: the pair registered before this one on the same list (same code)
  uint2* link_touched;     // (target, level) of every list that received a pair, in arrival order
  uint32_t* link_count;    // how many
};
size_t insert_lds_bytes(uint32_t ld, uint32_t ef);
hipError_t launch_insert_search(const InsertArgs& a, uint32_t n_new, hipStream_t st);
hipError_t launch_update_neigh(const InsertArgs& a, uint32_t n_items, const uint32_t* neigh, int level,
                               const uint32_t* cand_off, const uint32_t* cand_ids, hipStream_t st);
hipError_t launch_insert_link(const InsertArgs& a, uint32_t n_items, const uint32_t* tgt, const int32_t* tlevel,
                              const uint32_t* kind, const uint32_t* inc_off, const uint32_t* inc_ids, hipStream_t st);
hipError_t launch_insert_link_dev(const InsertArgs& a, uint32_t n_waves, hipStream_t st);

// k-way merge of per-shard (dist, id) result lists [n_lists][nq][k] -> [nq][k]; an id of list l enters as
// id * id_mul + l * id_step (row-sharded spaces: local row -> global row = local * G + shard)
hipError_t launch_merge_lists(const uint64_t* ids, const float* dist, const uint32_t* count, uint32_t nq,
                              uint32_t k, uint32_t n_lists, uint64_t* out_ids, float* out_dist,
                              uint32_t* out_count, hipStream_t st, size_t ids_stride, size_t dist_stride,
                              size_t count_stride, uint64_t id_mul = 1, uint64_t id_step = 0);

}  // namespace ehx
