// Candidate bookkeeping of the int8 filter pipeline (k_flati8.hip), 256 keys wide: one wave holds a sorted run of
// 256 (score, id) keys as 4 registers per lane (element e = r*64 + lane).
//   sample_select256_kernel  first thresholds from the sample pass's dumped lower bounds
//   select256_kernel         a pass's pool (unsorted) merged into the query's running best 256; publishes the
//                            k'-th best score as the next pass's threshold
//   rerank256_kernel         canonical (oracle-order) fp32 distances of the k' best lower bounds, top-k, certificate
#include "ehx_kernels.h"

namespace ehx {

namespace {

constexpr int kR = 4;  // registers per lane: 4 x 64 = 256 keys

__device__ __forceinline__ uint64_t umin64(uint64_t a, uint64_t b) { return a < b ? a : b; }
__device__ __forceinline__ uint64_t umax64(uint64_t a, uint64_t b) { return a < b ? b : a; }

// one compare-exchange layer of a bitonic network over the 256 keys: partner distance j, ascending where
// (e & size) == 0 (size 512 = everywhere ascending)
template <int SIZE, int J>
__device__ __forceinline__ void cx_layer(uint64_t (&k)[kR], int lane) {
  if (J >= 64) {
    constexpr int jr = J >> 6;
#pragma unroll
    for (int r = 0; r < kR; ++r) {
      if ((r & jr) == 0) {
        const int p = r | jr;
        const bool up = (((r << 6) | lane) & SIZE) == 0;
        const uint64_t lo = umin64(k[r], k[p]), hi = umax64(k[r], k[p]);
        k[r] = up ? lo : hi;
        k[p] = up ? hi : lo;
      }
    }
  } else {
#pragma unroll
    for (int r = 0; r < kR; ++r) {
      const uint64_t other = __shfl_xor(k[r], J, 64);
      const bool up = (((r << 6) | lane) & SIZE) == 0;
      const bool lower = (lane & J) == 0;
      const uint64_t lo = umin64(k[r], other), hi = umax64(k[r], other);
      k[r] = (lower == up) ? lo : hi;
    }
  }
}

template <int SIZE>
__device__ __forceinline__ void cx_merge(uint64_t (&k)[kR], int lane) {  // layers j = SIZE/2 .. 1
  if (SIZE >= 256) cx_layer<SIZE, 128>(k, lane);
  if (SIZE >= 128) cx_layer<SIZE, 64>(k, lane);
  if (SIZE >= 64) cx_layer<SIZE, 32>(k, lane);
  if (SIZE >= 32) cx_layer<SIZE, 16>(k, lane);
  if (SIZE >= 16) cx_layer<SIZE, 8>(k, lane);
  if (SIZE >= 8) cx_layer<SIZE, 4>(k, lane);
  if (SIZE >= 4) cx_layer<SIZE, 2>(k, lane);
  cx_layer<SIZE, 1>(k, lane);
}

// ascending bitonic sort of the 256 keys
__device__ __forceinline__ void wave_sort256(uint64_t (&k)[kR], int lane) {
  cx_merge<2>(k, lane);
  cx_merge<4>(k, lane);
  cx_merge<8>(k, lane);
  cx_merge<16>(k, lane);
  cx_merge<32>(k, lane);
  cx_merge<64>(k, lane);
  cx_merge<128>(k, lane);
  // last phase: everything ascending.  (e & 256) == 0 for every e < 256, so SIZE = 256 is "ascending everywhere".
  cx_merge<256>(k, lane);
}

// best (ascending) <- the 256 smallest of best U v, v ascending: reverse v, elementwise min (a bitonic
// sequence holding the 256 smallest), bitonic merge
__device__ __forceinline__ void wave_merge256(uint64_t (&best)[kR], const uint64_t (&v)[kR], int lane) {
#pragma unroll
  for (int r = 0; r < kR; ++r) {
    const uint64_t rv = __shfl(v[kR - 1 - r], 63 - lane, 64);  // element 255 - e
    best[r] = umin64(best[r], rv);
  }
  cx_merge<256>(best, lane);
}

// element `idx` (0..255) of a key run, broadcast to the wave
__device__ __forceinline__ uint64_t wave_pick256(const uint64_t (&k)[kR], uint32_t idx) {
  const uint32_t r = idx >> 6;
  uint64_t v = k[0];
  if (r == 1) v = k[1];
  if (r == 2) v = k[2];
  if (r == 3) v = k[3];
  return __shfl(v, (int)(idx & 63u), 64);
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// first thresholds: thr[q] = the `rank`-th smallest lower bound among the sample rows (rank <= 64); +inf when the
// sample holds fewer valid scores.  Any threshold is SOUND — the certificate (rerank256_kernel) is taken against
// the smallest threshold a query was ever scanned with (qparams.w, maintained by select256_kernel) — a low rank
// only bets that the final 256th best will still lie below it, which on a sample of 2048 rows is a safe bet
// (rank 16 is the 0.8 % quantile; the 256th best of even 100 k rows is the 0.26 % one).
// One workgroup of 4 waves per query: each wave keeps the best 64 of its quarter of the rows, wave 0 merges.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sample_select256_kernel(const float* __restrict__ scores, uint32_t n_rows,
                                                               uint32_t q_rows, uint32_t rank,
                                                               float* __restrict__ thr) {
  __shared__ uint64_t part[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const uint32_t q = blockIdx.x;
  auto sort64 = [&](uint64_t v) {
#pragma unroll
    for (int k2 = 2; k2 <= 64; k2 <<= 1) {
#pragma unroll
      for (int j = k2 >> 1; j > 0; j >>= 1) {
        const uint64_t other = __shfl_xor(v, j, 64);
        const bool up = (lane & k2) == 0;
        const bool lower = (lane & j) == 0;
        const uint64_t lo = umin64(v, other), hi = umax64(v, other);
        v = (lower == up) ? lo : hi;
      }
    }
    return v;
  };
  auto merge64 = [&](uint64_t best, uint64_t v_sorted) {
    const uint64_t rv = __shfl(v_sorted, 63 - lane, 64);
    best = umin64(best, rv);
#pragma unroll
    for (int j = 32; j > 0; j >>= 1) {
      const uint64_t other = __shfl_xor(best, j, 64);
      const uint64_t lo = umin64(best, other), hi = umax64(best, other);
      best = (lane & j) == 0 ? lo : hi;
    }
    return best;
  };
  uint64_t best = kKeyInf;
  for (uint32_t r0 = (uint32_t)w * 64u; r0 < n_rows; r0 += 256) {
    const uint32_t row = r0 + (uint32_t)lane;
    uint64_t key = kKeyInf;
    if (row < n_rows) {
      const float sc = scores[(size_t)row * q_rows + q];
      if (sc == sc && sc < __builtin_inff()) key = ((uint64_t)f32_to_ordered(sc) << 32) | 0xFFFFFFFFull;
    }
    best = merge64(best, sort64(key));
  }
  part[w][lane] = best;
  __syncthreads();
  if (w == 0) {
    best = merge64(best, part[1][lane]);
    best = merge64(best, part[2][lane]);
    best = merge64(best, part[3][lane]);
    const uint64_t kth = __shfl(best, (int)(rank - 1), 64);
    if (lane == 0) thr[q] = kth == kKeyInf ? __builtin_inff() : ordered_to_f32((uint32_t)(kth >> 32));
  }
}

hipError_t launch_sample_select256(const float* scores, uint32_t n_rows, uint32_t q_rows, uint32_t nq,
                                   uint32_t rank, float* thr, hipStream_t st) {
  if (rank < 1) rank = 1;
  if (rank > 64) rank = 64;
  hipLaunchKernelGGL(sample_select256_kernel, dim3(nq), dim3(256), 0, st, scores, n_rows, q_rows, rank, thr);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// one pass's pool -> running best 256 of the query.  Every key the scan collected is <= the pass's threshold;
// everything it did not collect is above it, so the kprime-th best after the merge is again an upper bound of the
// final kprime-th best and serves as the next pass's threshold.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void select256_kernel(const uint64_t* __restrict__ pool, uint32_t* __restrict__ pool_cnt,
                                                       uint32_t pool_cap, uint32_t kprime, uint64_t* __restrict__ merged,
                                                       uint32_t seed, float* __restrict__ thr,
                                                       float4* __restrict__ qparams) {
  const int lane = threadIdx.x;
  const uint32_t q = blockIdx.x;
  uint64_t best[kR];
#pragma unroll
  for (int r = 0; r < kR; ++r) best[r] = seed ? merged[(size_t)q * kMerged8 + r * 64 + lane] : kKeyInf;
  uint32_t n = pool_cnt[q];
  if (n > pool_cap) n = pool_cap;  // (overflowed pool: the query is flagged; keep what fits)
  const uint64_t* p = pool + (size_t)q * pool_cap;
  for (uint32_t b0 = 0; b0 < n; b0 += 256) {
    uint64_t v[kR];
#pragma unroll
    for (int r = 0; r < kR; ++r) {
      const uint32_t i = b0 + (uint32_t)r * 64u + (uint32_t)lane;
      v[r] = i < n ? p[i] : kKeyInf;
    }
    // a chunk in which nothing beats the 256th best so far cannot change anything
    const uint64_t bar = wave_pick256(best, 255);
    const uint64_t vmin = umin64(umin64(v[0], v[1]), umin64(v[2], v[3]));
    if (!__any(vmin < bar)) continue;
    wave_sort256(v, lane);
    wave_merge256(best, v, lane);
  }
#pragma unroll
  for (int r = 0; r < kR; ++r) merged[(size_t)q * kMerged8 + r * 64 + lane] = best[r];
  const uint64_t kth = wave_pick256(best, kprime - 1);
  if (lane == 0) {
    // qparams.w = the smallest threshold this query was ever scanned with: every row that is in no pool had a
    // lower bound above it (thr[q] still holds the threshold of the pass just merged)
    const float used = thr[q];
    if (used < qparams[q].w) qparams[q].w = used;
    // Thresholds only ever tighten.  The list is complete up to the threshold just used and no further (later
    // passes never collected what lay above it), so an entry of the requested rank that lies ABOVE it comes from
    // the incomplete zone and can be far looser than the rank suggests — enough, at a low rank, to overflow the
    // next pass's pool (simulated and measured: 0.2 % of the queries at 10 M rows).
    const float cand = kth == kKeyInf ? __builtin_inff() : ordered_to_f32((uint32_t)(kth >> 32));
    thr[q] = fminf(used, cand);
    pool_cnt[q] = 0u;
  }
}

hipError_t launch_select256(const uint64_t* pool, uint32_t* pool_cnt, uint32_t pool_cap, uint32_t nq, uint32_t kprime,
                            uint64_t* merged, bool seed, float* thr, float4* qparams, hipStream_t st) {
  hipLaunchKernelGGL(select256_kernel, dim3(nq), dim3(64), 0, st, pool, pool_cnt, pool_cap, kprime, merged,
                     seed ? 1u : 0u, thr, qparams);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// canonical re-rank of the int8 filter's candidates.  merged[0..kprime) are the query's best lower bounds in
// ascending order.  They are evaluated in that order, 64 per round, in the oracle's summation order, the wave
// keeping the best 64 exact (distance, id) keys so far.
// Certificate.  A row that is NOT in the list either was never collected — its lower bound was above the threshold
// of the pass that scanned it, hence above the smallest threshold the query was ever scanned with (qparams.w) — or
// was dropped by a merge, hence is not below the list's last entry (if the list is full).  So with
//     floor = min(qparams.w, last entry of a full list)
// every outsider's lower bound is >= floor, and every not-yet-evaluated candidate's is >= the next candidate's.
// After a round:  D(min(floor, next candidate)) - margin > exact k-th distance so far  proves that nothing unseen
// can enter the top-k: the query is certified and the remaining candidates are never read.  Consuming the whole
// list leaves the same test against the floor alone.
// ---------------------------------------------------------------------------------------------
namespace {
struct RerankState {
  uint64_t best;   // ascending best-64 exact keys so far (one per lane)
  bool certified;
};

// one round's 64 exact keys (one per lane, unsorted) -> merged into st.best; then the early-stop test against
// candidate `nxt` (index into the list) and the floor
__device__ __forceinline__ bool rerank_round(const Rerank256Args& a, const uint64_t* mq, uint64_t key, uint32_t nxt,
                                             float floor_s, const float2 uv, float qn, float maxss, int lane,
                                             RerankState& st) {
  uint64_t v = key;
#pragma unroll
  for (int k2 = 2; k2 <= 64; k2 <<= 1) {
#pragma unroll
    for (int j = k2 >> 1; j > 0; j >>= 1) {
      const uint64_t other = __shfl_xor(v, j, 64);
      const bool up = (lane & k2) == 0;
      const bool lower = (lane & j) == 0;
      const uint64_t lo = umin64(v, other), hi = umax64(v, other);
      v = (lower == up) ? lo : hi;
    }
  }
  const uint64_t rv = __shfl(v, 63 - lane, 64);
  st.best = umin64(st.best, rv);
#pragma unroll
  for (int j = 32; j > 0; j >>= 1) {
    const uint64_t other = __shfl_xor(st.best, j, 64);
    const uint64_t lo = umin64(st.best, other), hi = umax64(st.best, other);
    st.best = (lane & j) == 0 ? lo : hi;
  }
  if (a.k == 0) return false;
  const uint64_t kk = __shfl(st.best, (int)a.k - 1, 64);
  if (kk == kKeyInf) return false;
  float s_next = floor_s;
  if (nxt < a.kprime) {
    const uint64_t nk = mq[nxt];
    if (nk != kKeyInf) s_next = fminf(s_next, ordered_to_f32((uint32_t)(nk >> 32)));
  }
  const float kth = ordered_to_f32((uint32_t)(kk >> 32));
  const float lb = __builtin_fmaf(uv.x, s_next, uv.y);
  const float margin = cert_margin(a.metric, a.dims, qn, maxss, fmaxf(fabsf(kth), fabsf(lb)));
  return lb - margin > kth;  // (NaN u / v: a query the filter could not bound -> never certified)
}

__device__ __forceinline__ void rerank_finish(const Rerank256Args& a, uint32_t q, const RerankState& st, int lane) {
  const uint32_t nvalid = (uint32_t)__builtin_popcountll(__ballot(st.best != kKeyInf));
  const uint32_t cnt = nvalid < a.k ? nvalid : a.k;
  if (lane < (int)a.k) {
    const bool ok = (uint32_t)lane < cnt;
    a.out_ids[(size_t)q * a.k + lane] = ok ? (uint64_t)(uint32_t)st.best : ~0ull;
    a.out_dist[(size_t)q * a.k + lane] = ok ? ordered_to_f32((uint32_t)(st.best >> 32)) : __builtin_inff();
  }
  bool uncert;
  if (a.ovf[q]) uncert = true;                        // a pool overflowed in some pass: candidates may be missing
  else if (cnt < a.k || a.k == 0) uncert = a.n > cnt;  // candidates lost (NaN rows / queries) or fewer rows than k
  else uncert = !st.certified;
  if (lane == 0) {
    a.out_count[q] = cnt;
    if (uncert) atomicAdd(a.n_uncertified, 1ull);
    if (a.uncert_flags) a.uncert_flags[q] = uncert ? 1u : 0u;
  }
}

__device__ __forceinline__ float rerank_floor(const Rerank256Args& a, uint32_t q, const uint64_t* mq) {
  float floor_s = a.qparams[q].w;
  const uint64_t last = mq[a.kprime - 1];
  if (last != kKeyInf) floor_s = fminf(floor_s, ordered_to_f32((uint32_t)(last >> 32)));
  return floor_s;
}
}  // namespace

// fp32 rows: one wave per query, one candidate per lane — the lane walks its row with 16-byte loads through a
// register ring (canon_dist_lane_t: 16-24 loads in flight per lane), the query sits in LDS
template <int METRIC01, bool SCALE>
__global__ __launch_bounds__(64) void rerank256_lane_kernel(const Rerank256Args a, uint32_t q_in_lds) {
  extern __shared__ __attribute__((aligned(16))) float qs[];
  const int lane = threadIdx.x;
  const uint32_t q = blockIdx.x;
  const float* qv = a.Q + (size_t)q * a.ld;
  if (q_in_lds) {
    for (uint32_t i = lane; i < a.ld; i += 64) qs[i] = qv[i];
    __syncthreads();
  }
  const uint64_t* mq = a.merged + (size_t)q * kMerged8;
  const float2 uv = a.quv[q];
  const float qn = a.metric == 0 ? uv.y : (a.metric == 1 ? uv.x * uv.x : 1.0f);
  const float maxss = a.max_sumsq ? *a.max_sumsq : __builtin_inff();
  const float floor_s = rerank_floor(a, q, mq);
  RerankState st{kKeyInf, false};
  for (uint32_t c0 = 0; c0 < a.kprime; c0 += 64) {
    const uint32_t ci = c0 + (uint32_t)lane;
    const uint64_t mk = ci < a.kprime ? mq[ci] : kKeyInf;
    const uint32_t id = (uint32_t)mk;
    const bool valid = mk != kKeyInf && id < a.n;
    float d = __builtin_inff();
    if (valid) {
      const float* xv = (const float*)a.X + (size_t)id * a.ld;
      const float xs = SCALE ? a.inv_norm[id] : 1.0f;
      d = q_in_lds ? canon_dist_lane_t<METRIC01, SCALE>(qs, xv, xs, a.dims)
                   : canon_dist_lane_t<METRIC01, SCALE>(qv, xv, xs, a.dims);
    }
    // (a NaN distance — a row or query holding NaN — is never a neighbour: the key is dropped)
    const uint64_t key = (valid && d == d) ? (((uint64_t)f32_to_ordered(d) << 32) | id) : kKeyInf;
    st.certified = rerank_round(a, mq, key, c0 + 64, floor_s, uv, qn, maxss, lane, st);
    if (st.certified) break;
  }
  rerank_finish(a, q, st, lane);
}

// fp16 rows: four lanes per candidate (canon_dist widens the halves exactly), 64 candidates per round per workgroup
template <typename XT>
__global__ __launch_bounds__(256) void rerank256_kernel(const Rerank256Args a) {
  __shared__ uint64_t keys[64];
  __shared__ int stop_flag;
  const int tid = threadIdx.x;
  const uint32_t q = blockIdx.x;
  const int g = tid >> 2, sub = tid & 3;
  const float* qv = a.Q + (size_t)q * a.ld;
  const bool scale_x = a.metric == 2;
  const uint64_t* mq = a.merged + (size_t)q * kMerged8;
  const float2 uv = a.quv[q];
  const float qn = a.metric == 0 ? uv.y : (a.metric == 1 ? uv.x * uv.x : 1.0f);
  const float maxss = a.max_sumsq ? *a.max_sumsq : __builtin_inff();
  const float floor_s = rerank_floor(a, q, mq);
  RerankState st{kKeyInf, false};
  if (tid == 0) stop_flag = 0;
  __syncthreads();
  for (uint32_t c0 = 0; c0 < a.kprime; c0 += 64) {
    const uint32_t ci = c0 + (uint32_t)g;
    const uint64_t mk = ci < a.kprime ? mq[ci] : kKeyInf;
    const uint32_t id = (uint32_t)mk;
    const bool valid = mk != kKeyInf && id < a.n;
    float d = __builtin_inff();
    if (valid) {
      const XT* xv = (const XT*)a.X + (size_t)id * a.ld;
      const float xs = scale_x ? a.inv_norm[id] : 1.0f;
      d = canon_dist(a.metric == 0 ? 0 : 1, qv, xv, xs, scale_x, a.dims, sub);
    }
    if (sub == 0) keys[g] = (valid && d == d) ? (((uint64_t)f32_to_ordered(d) << 32) | id) : kKeyInf;
    __syncthreads();
    if (tid < 64) {
      st.certified = rerank_round(a, mq, keys[tid], c0 + 64, floor_s, uv, qn, maxss, tid, st);
      if (st.certified && tid == 0) stop_flag = 1;
    }
    __syncthreads();
    if (stop_flag) break;
  }
  if (tid < 64) rerank_finish(a, q, st, tid);
}

hipError_t launch_rerank256(const Rerank256Args& a, hipStream_t st) {
  if (a.x_half) {
    hipLaunchKernelGGL(rerank256_kernel<__half>, dim3(a.nq), dim3(256), 0, st, a);
    return hipGetLastError();
  }
  const size_t qbytes = (size_t)a.ld * sizeof(float);
  const uint32_t in_lds = qbytes <= 48 * 1024 ? 1u : 0u;  // (very long rows: the query stays in global memory)
  const size_t lds = in_lds ? qbytes : 0;
  if (a.metric == 0) hipLaunchKernelGGL((rerank256_lane_kernel<0, false>), dim3(a.nq), dim3(64), lds, st, a, in_lds);
  else if (a.metric == 1) hipLaunchKernelGGL((rerank256_lane_kernel<1, false>), dim3(a.nq), dim3(64), lds, st, a, in_lds);
  else hipLaunchKernelGGL((rerank256_lane_kernel<1, true>), dim3(a.nq), dim3(64), lds, st, a, in_lds);
  return hipGetLastError();
}

}  // namespace ehx
