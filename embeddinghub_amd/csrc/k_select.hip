// Candidate bookkeeping of the int8 filter pipeline (k_flati8.hip), 256 keys wide: one wave holds a sorted run of
// 256 (score, id) keys as 4 registers per lane (element e = r*64 + lane).
//   sample_select256_kernel  first thresholds from the sample pass's dumped lower bounds
//   select256_kernel         a pass's pool (unsorted) merged into the query's running best 256; publishes the
//                            k'-th best score as the next pass's threshold
//   rerank256_kernel         canonical (oracle-order) fp32 distances of the k' best lower bounds, top-k, certificate
#include "ehx_env.h"
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
//
// Round 6: ONE wave per query, no sorting at all.  The sample pass dumps its scores in 16 x 16 blocks (scan8_dump_index: a
// query's scores come as 64-byte runs, read with eight float4 loads per lane), every lane keeps its 32 scores as
// order-preserving 32-bit keys in registers, and the rank-th smallest key is found by bisection of the key range: a step
// counts the keys <= mid with one compare + ballot + scalar popcount per register (no cross-lane data movement), ~25 steps
// for scores that span a few binades.  Rounds 2-5: four waves per query, each sorting its quarter 64 keys at a time with
// 64-bit bitonic networks (28 shuffle layers per 64 scores) from a row-major dump read one 4-byte element per 4-KiB
// stride — 25-27 us per batch whatever the shape, 2.6 % of a 6.25 M x 128 batch; now ~6.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void sample_select256_kernel(const float* __restrict__ scores, uint32_t n_rows,
                                                              uint32_t nq, uint32_t rank, float* __restrict__ thr) {
  constexpr int kV = 8;  // float4 loads per lane: up to 8 x 64 x 4 = 2048 scores
  const int lane = threadIdx.x;
  const uint32_t q = blockIdx.x;
  if (q >= nq) return;
  const uint32_t n4 = n_rows >> 2;  // (n_rows % 16 == 0: the sample is whole tiles)
  uint32_t k[kV * 4];
  uint32_t kmin = 0xFFFFFFFFu, kmax = 0u, n_valid = 0u;
#pragma unroll
  for (int i = 0; i < kV; ++i) {
    const uint32_t at = (uint32_t)i * 64u + (uint32_t)lane;
    float4 v = make_float4(__builtin_inff(), __builtin_inff(), __builtin_inff(), __builtin_inff());
    if (at < n4) v = *(const float4*)(scores + scan8_dump_index(q, at << 2, n_rows));   // rows 4 at .. 4 at + 3: 16 bytes of a 64-byte run
    const float f[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool ok = f[j] == f[j] && f[j] < __builtin_inff();
      const uint32_t key = ok ? f32_to_ordered(f[j]) : 0xFFFFFFFFu;
      k[i * 4 + j] = key;
      kmin = min(kmin, key);
      kmax = ok ? max(kmax, key) : kmax;
      n_valid += (uint32_t)__builtin_popcountll(__ballot(ok));   // (wave-uniform)
    }
  }
  if (n_valid < rank) {   // fewer valid scores than the rank asks for
    if (lane == 0) thr[q] = __builtin_inff();
    return;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    kmin = min(kmin, (uint32_t)__shfl_xor((int)kmin, off, 64));
    kmax = max(kmax, (uint32_t)__shfl_xor((int)kmax, off, 64));
  }
  // smallest key x with |{keys <= x}| >= rank, x in [kmin, kmax] (kmax qualifies: n_valid >= rank)
  uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)kmin), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)kmax);
  while (lo < hi) {
    const uint32_t mid = lo + ((hi - lo) >> 1);
    uint32_t cnt = 0u;
#pragma unroll
    for (int i = 0; i < kV * 4; ++i) cnt += (uint32_t)__builtin_popcountll(__ballot(k[i] <= mid));
    if (cnt >= rank) hi = mid;
    else lo = mid + 1u;
  }
  if (lane == 0) thr[q] = ordered_to_f32(lo);
}

hipError_t launch_sample_select256(const float* scores, uint32_t n_rows, uint32_t q_rows, uint32_t nq,
                                   uint32_t rank, float* thr, hipStream_t st) {
  (void)q_rows;
  if (rank < 1) rank = 1;
  if (rank > 64) rank = 64;
  if (n_rows == 0 || n_rows > 2048 || (n_rows & 15u)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(sample_select256_kernel, dim3(nq), dim3(64), 0, st, scores, n_rows, nq, rank, thr);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// one pass's pool -> running best 256 of the query.  Every key the scan collected is <= the pass's threshold;
// everything it did not collect is above it, so the kprime-th best after the merge is again an upper bound of the
// final kprime-th best and serves as the next pass's threshold.
//
// One workgroup of four waves per query; 256 pool keys per trip.  Wave w sorts its 64 keys (one register per lane);
// then every key — the 256 of the running list and the 256 new ones — finds its place in the merged order by
// binary searches over the OTHER sorted runs in LDS (keys are distinct: the row id is part of the key), and is
// written there if the place is among the first 256.  Round 2 kept the list in one wave's registers and sorted each
// chunk of 256 with a 36-layer bitonic network of 64-bit cross-lane exchanges: ~50 us per launch whatever the pool
// held (three to four launches per batch: a seventh of a 1.25 M-row shard's batch).
// ---------------------------------------------------------------------------------------------
namespace {
__device__ __forceinline__ uint32_t lower_bound_lds(const uint64_t* a, uint32_t n, uint64_t key) {  // n a power of two
  uint32_t lo = 0;
  for (uint32_t step = n >> 1; step > 0; step >>= 1)
    if (a[lo + step - 1] < key) lo += step;
  return lo + (a[lo] < key ? 1u : 0u);
}
}  // namespace

__global__ __launch_bounds__(256) void select256_kernel(const uint64_t* __restrict__ pool, uint32_t* __restrict__ pool_cnt,
                                                        uint32_t pool_cap, uint32_t kprime, uint64_t* __restrict__ merged,
                                                        uint32_t width, uint32_t seed, float* __restrict__ thr,
                                                        float4* __restrict__ qparams) {
  __shared__ uint64_t best[2][kMerged8Max];  // the running list (ascending, `width` long), ping-pong
  __shared__ uint64_t runs[4][64];        // this trip's new keys: four sorted runs
  __shared__ int any_new;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const uint32_t q = blockIdx.x;
  uint32_t cur = 0;
  for (uint32_t e = (uint32_t)tid; e < width; e += 256) best[0][e] = seed ? merged[(size_t)q * width + e] : kKeyInf;
  uint32_t n = pool_cnt[q];
  if (n > pool_cap) n = pool_cap;  // (overflowed pool: the query is flagged; keep what fits)
  const uint64_t* p = pool + (size_t)q * pool_cap;
  for (uint32_t b0 = 0; b0 < n; b0 += 256) {
    if (tid == 0) any_new = 0;
    __syncthreads();  // best[cur] complete, flag cleared
    const uint32_t i = b0 + (uint32_t)tid;
    uint64_t v = i < n ? p[i] : kKeyInf;
    // a chunk in which nothing beats the 256th best so far cannot change anything
    if (v < best[cur][width - 1]) any_new = 1;
    __syncthreads();
    if (!any_new) continue;
    // ascending bitonic sort of the wave's 64 keys
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
    runs[w][lane] = v;
    const uint32_t nxt = cur ^ 1u;
    for (uint32_t e = (uint32_t)tid; e < width; e += 256) best[nxt][e] = kKeyInf;  // (places nobody claims stay empty)
    __syncthreads();
    // the new key's place: its index in its own run + the keys below it in the other runs and in the list
    if (v != kKeyInf) {
      uint32_t pos = (uint32_t)lane + lower_bound_lds(best[cur], width, v);
#pragma unroll
      for (int o = 0; o < 4; ++o)
        if (o != w) pos += lower_bound_lds(runs[o], 64, v);
      if (pos < width) best[nxt][pos] = v;
    }
    // the old keys' places
    for (uint32_t e = (uint32_t)tid; e < width; e += 256) {
      const uint64_t old = best[cur][e];
      if (old == kKeyInf) continue;
      uint32_t pos = e;
#pragma unroll
      for (int o = 0; o < 4; ++o) pos += lower_bound_lds(runs[o], 64, old);
      if (pos < width) best[nxt][pos] = old;
    }
    cur = nxt;
  }
  __syncthreads();
  for (uint32_t e = (uint32_t)tid; e < width; e += 256) merged[(size_t)q * width + e] = best[cur][e];
  if (tid == 0) {
    const uint64_t kth = best[cur][kprime - 1];
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
                            uint64_t* merged, uint32_t width, bool seed, float* thr, float4* qparams, hipStream_t st) {
  if (width < 256 || width > kMerged8Max || (width & (width - 1)) || kprime < 1 || kprime > width) return hipErrorInvalidValue;
  hipLaunchKernelGGL(select256_kernel, dim3(nq), dim3(256), 0, st, pool, pool_cnt, pool_cap, kprime, merged, width,
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
    if (a.uncert_flags) {
      // 2 = the candidate LIST was what failed: it is full and its last entry — not the scan's threshold — is the floor
      // the certificate ran into, so a wider list may certify this query (the host widens the space's list on these
      // only: a pool overflow, lost candidates or a threshold-bound floor are not cured by width); 1 = any other cause
      const uint64_t last = a.merged[(size_t)q * a.width + a.kprime - 1];
      const bool by_list = last != kKeyInf && ordered_to_f32((uint32_t)(last >> 32)) <= a.qparams[q].w;
      a.uncert_flags[q] = !uncert ? 0u : ((!a.ovf[q] && cnt >= a.k && a.k != 0 && by_list) ? 2u : 1u);
    }
  }
}

__device__ __forceinline__ float rerank_floor(const Rerank256Args& a, uint32_t q, const uint64_t* mq) {
  float floor_s = a.qparams[q].w;
  const uint64_t last = mq[a.kprime - 1];
  if (last != kKeyInf) floor_s = fminf(floor_s, ordered_to_f32((uint32_t)(last >> 32)));
  return floor_s;
}
}  // namespace

// Canonical distances of 64 rows (one per lane: row `id`, garbage allowed when !valid) to the query q, for rows whose
// length is a multiple of 32 floats.  The arithmetic is the lane-private walk's (canon_lane_step, piece after piece);
// what changes is how the rows reach the lane.  A lane that loads its own row 16 bytes at a time makes every wave
// load touch 64 different 128-byte lines, each needed eight times: with three ring blocks in flight per lane and four
// waves on a CU that is 768 lines against a 256-line L1, and most of them are evicted before their eighth use
// (measured: 0.41 ms for 1024 queries x 64 rows x 3 KB = 0.5 TB/s).  Here the wave loads the rows TOGETHER —
// eight lanes per row take its next 128-byte line in one request, eight rows per load instruction — parks the
// pieces in LDS and every lane then reads its own row's eight pieces back.  Piece p of row r sits at slot
// p ^ ((r >> 1) & 7) of the row's line: the write is contiguous per instruction, the read conflict-free.  One chunk
// (128 bytes of every row) is in flight while the previous one is accumulated.
//
// HALFX (round 6): binary16 rows — a 128-byte line is 64 elements, a 16-byte piece eight of them, widened exactly and fed to
// the same steps as two float4 (rows of a multiple of 64 elements).  Until then fp16 spaces re-ranked with four lanes per
// candidate walking their row in 8-byte pieces (rerank256_kernel<__half>): 2.03 ms per batch at 12.5 M x 1536 with a
// 1024-wide list — 3-6 x what the bytes cost; it still serves rows of other lengths.
constexpr uint32_t kStageFloat4 = 64 * 8;  // 64 rows x 8 pieces (8 KB)
typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
template <int METRIC01, bool SCALE, bool HALFX = false>
__device__ __forceinline__ float wave_rows_dist_staged(const float* __restrict__ q_lds, const void* __restrict__ X,
                                                       uint32_t ld, uint32_t dims, uint32_t id, bool valid, float xscale,
                                                       float4* stage, int lane) {
  const uint32_t n_chunks = dims / (HALFX ? 64u : 32u);
  const size_t row_bytes = (size_t)ld * (HALFX ? 2u : 4u);
  const int sub = lane & 7, grp = lane >> 3;
  const uint32_t my = valid ? id : 0u;
  // the row this lane helps to load in instruction i is row i*8 + grp of the wave's 64
#define EHX_SRC(i)                                                                                               \
  ((const float4*)((const char*)X + (size_t)(uint32_t)__shfl((int)my, (i) * 8 + grp, 64) * row_bytes) + \
   (sub ^ ((((i) * 8 + grp) >> 1) & 7)))
  const float4 *s0 = EHX_SRC(0), *s1 = EHX_SRC(1), *s2 = EHX_SRC(2), *s3 = EHX_SRC(3), *s4 = EHX_SRC(4),
               *s5 = EHX_SRC(5), *s6 = EHX_SRC(6), *s7 = EHX_SRC(7);
#undef EHX_SRC
  const int key = (lane >> 1) & 7;
  const float4* q4 = (const float4*)q_lds;
  float p0 = 0.0f, p1 = 0.0f, p2 = 0.0f, p3 = 0.0f;
  // Three chunks (128 bytes of every row each) in flight: with one chunk ahead — round 2 — the wave, alone on its SIMD,
  // paid most of an HBM round trip per chunk (24 chunks of a 768-dim row: 0.09 ms per batch, the re-rank's whole time).
  float4 a0, a1, a2, a3, a4, a5, a6, a7, b0, b1, b2, b3, b4, b5, b6, b7, c0, c1, c2, c3, c4, c5, c6, c7;
#define EHX_FETCH(R, C)                                                                                  \
  do {                                                                                                   \
    if ((C) < n_chunks) {                                                                                \
      const size_t o_ = (size_t)(C) * 8;                                                                 \
      R##0 = s0[o_], R##1 = s1[o_], R##2 = s2[o_], R##3 = s3[o_], R##4 = s4[o_], R##5 = s5[o_], R##6 = s6[o_], \
      R##7 = s7[o_];                                                                                     \
    }                                                                                                    \
  } while (0)
  // (one staging buffer is enough: the LDS executes a wave's accesses in order, so a chunk's writes land after the
  // previous chunk's reads); the register block just parked is free again: the chunk three ahead is requested into it
#define EHX_CONSUME(R, C)                                                                                \
  do {                                                                                                   \
    wave_lds_sync();                                                                                     \
    stage[0 * 64 + lane] = R##0, stage[1 * 64 + lane] = R##1, stage[2 * 64 + lane] = R##2;               \
    stage[3 * 64 + lane] = R##3, stage[4 * 64 + lane] = R##4, stage[5 * 64 + lane] = R##5;               \
    stage[6 * 64 + lane] = R##6, stage[7 * 64 + lane] = R##7;                                            \
    EHX_FETCH(R, (C) + 3);                                                                               \
    wave_lds_sync();                                                                                     \
    _Pragma("unroll") for (int p = 0; p < 8; ++p) {                                                      \
      const float4 raw_ = stage[lane * 8 + (p ^ key)];                                                   \
      if constexpr (HALFX) {                                                                             \
        const h8_t h_ = __builtin_bit_cast(h8_t, raw_);                                                  \
        canon_lane_step<METRIC01, SCALE>(make_float4((float)h_[0], (float)h_[1], (float)h_[2], (float)h_[3]), \
                                         q4[(size_t)(C) * 16 + 2 * p], xscale, p0, p1, p2, p3);          \
        canon_lane_step<METRIC01, SCALE>(make_float4((float)h_[4], (float)h_[5], (float)h_[6], (float)h_[7]), \
                                         q4[(size_t)(C) * 16 + 2 * p + 1], xscale, p0, p1, p2, p3);      \
      } else {                                                                                           \
        canon_lane_step<METRIC01, SCALE>(raw_, q4[(size_t)(C) * 8 + p], xscale, p0, p1, p2, p3);         \
      }                                                                                                  \
    }                                                                                                    \
  } while (0)
  a0 = a1 = a2 = a3 = a4 = a5 = a6 = a7 = b0 = b1 = b2 = b3 = b4 = b5 = b6 = b7 = c0 = c1 = c2 = c3 = c4 = c5 = c6 = c7 =
      make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  EHX_FETCH(a, 0u);
  EHX_FETCH(b, 1u);
  EHX_FETCH(c, 2u);
  for (uint32_t ch = 0; ch < n_chunks; ch += 3) {
    EHX_CONSUME(a, ch);
    if (ch + 1 < n_chunks) EHX_CONSUME(b, ch + 1);
    if (ch + 2 < n_chunks) EHX_CONSUME(c, ch + 2);
  }
#undef EHX_CONSUME
#undef EHX_FETCH
  float res = ex_add(ex_add(ex_add(p0, p1), p2), p3);
  if (METRIC01 != 0) res = ex_sub(1.0f, res);
  return res;
}

// fp32 rows: one wave per query, one candidate per lane; rows of a multiple of 32 floats are loaded by the wave
// together (wave_rows_dist_staged), others by the lane itself with 16-byte loads through a register ring
// (canon_dist_lane_t); the query sits in LDS
template <int METRIC01, bool SCALE, bool STAGED, bool HALFX = false>
__global__ __launch_bounds__(64) void rerank256_lane_kernel(const Rerank256Args a, uint32_t q_in_lds) {
  static_assert(!HALFX || STAGED, "binary16 rows: the staged form only");
  extern __shared__ __attribute__((aligned(16))) float qs[];
  float4* stage = (float4*)(qs + (q_in_lds ? a.ld : 0u));  // [kStageFloat4] when `staged`
  const int lane = threadIdx.x;
  const uint32_t q = blockIdx.x;
  const float* qv = a.Q + (size_t)q * a.ld;
  if (q_in_lds) {
    for (uint32_t i = lane; i < a.ld; i += 64) qs[i] = qv[i];
    __syncthreads();
  }
  const uint64_t* mq = a.merged + (size_t)q * a.width;
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
    if (STAGED) {
      const float xs = (SCALE && valid) ? a.inv_norm[id] : 1.0f;
      d = wave_rows_dist_staged<METRIC01, SCALE, HALFX>(qs, a.X, a.ld, a.dims, id, valid, xs, stage, lane);
    } else if (!HALFX && valid) {
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
  const uint64_t* mq = a.merged + (size_t)q * a.width;
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
  const size_t qbytes = (size_t)a.ld * sizeof(float);
  const uint32_t in_lds = qbytes <= 32 * 1024 ? 1u : 0u;  // (very long rows: the query stays in global memory)
  const bool allow_staged = env().rerank_staged;  // (EHX_RERANK_STAGED=0: every lane walks its own row, A/B runs)
  if (a.x_half) {
    // binary16 rows of a multiple of 64 elements: one wave per query, the rows loaded by the wave together (round 6)
    if (allow_staged && in_lds && (a.dims & 63u) == 0 && a.dims >= 64) {
      const size_t ldsh = qbytes + kStageFloat4 * sizeof(float4);
      if (a.metric == 0) hipLaunchKernelGGL((rerank256_lane_kernel<0, false, true, true>), dim3(a.nq), dim3(64), ldsh, st, a, 1u);
      else if (a.metric == 1) hipLaunchKernelGGL((rerank256_lane_kernel<1, false, true, true>), dim3(a.nq), dim3(64), ldsh, st, a, 1u);
      else hipLaunchKernelGGL((rerank256_lane_kernel<1, true, true, true>), dim3(a.nq), dim3(64), ldsh, st, a, 1u);
      return hipGetLastError();
    }
    hipLaunchKernelGGL(rerank256_kernel<__half>, dim3(a.nq), dim3(256), 0, st, a);
    return hipGetLastError();
  }
  const uint32_t staged = (allow_staged && in_lds && (a.dims & 31u) == 0 && a.dims >= 32) ? 1u : 0u;
  const size_t lds = (in_lds ? qbytes : 0) + (staged ? kStageFloat4 * sizeof(float4) : 0);
#define EHX_RR(M, S)                                                                                              \
  do {                                                                                                            \
    if (staged) hipLaunchKernelGGL((rerank256_lane_kernel<M, S, true>), dim3(a.nq), dim3(64), lds, st, a, in_lds); \
    else hipLaunchKernelGGL((rerank256_lane_kernel<M, S, false>), dim3(a.nq), dim3(64), lds, st, a, in_lds);       \
  } while (0)
  if (a.metric == 0) EHX_RR(0, false);
  else if (a.metric == 1) EHX_RR(1, false);
  else EHX_RR(1, true);
#undef EHX_RR
  return hipGetLastError();
}

}  // namespace ehx
