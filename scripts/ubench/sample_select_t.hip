// Microbenchmark + correctness check for the first-threshold selection of the int8 engine (k_select.hip,
// sample_select256_kernel: thr[q] = the rank-th smallest of scores[0..n_rows)[q], rank <= 64, +inf if fewer valid).
// The shipped kernel gives every query its own workgroup, so each lane reads a different row of the sample scores —
// 4 bytes out of every 4-KB-strided line: 0.027 ms per batch of 1024 queries, 2.5 % of a 1 M-row batch.  The candidate
// here turns the access around: a workgroup owns 64 CONSECUTIVE queries, lane = query, so a wave reads one row's 64
// scores as one 256-byte piece; the four waves split the rows and every lane keeps the best `rank` of its query in a
// sorted list in LDS (a new score is compared with the list's last entry first: after a few dozen rows almost every
// score is rejected by that one compare), then lane q of wave 0 merges the four lists.
//
//   hipcc --offload-arch=gfx950 -O3 -o sample_select_t.bin scripts/ubench/sample_select_t.hip && ./sample_select_t.bin
//
// RESULT (MI355X, profiles/r03_sample_select_transposed_ubench.txt): both kernels agree with the host on all 1000 queries,
// and the candidate LOSES by an order of magnitude — shipped 28 us per launch at every rank, transposed 301 / 637 /
// 2527 us at rank 8 / 16 / 64: only 16 workgroups, and a wave's 64 private insertion loops serialise (some lane
// inserts on nearly every row, and every step of its shift is a dependent LDS round trip for the whole wave).  The
// shipped kernel's uncoalesced reads are cheaper than that; a faster variant would have to keep the shuffle-network
// selection and only stage the scores through an LDS transpose.  Kept as the record of the measurement.
//
// Prints, for rank 8 / 16 / 64 on 2048 x 1024 scores: microseconds per launch of both kernels (HIP events, 200
// launches) and whether both agree with the host's nth_element on every query (NaN / +inf scores, which the scan
// writes for padding rows, are not candidates).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                  \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); \
      exit(1);                                                                    \
    }                                                                             \
  } while (0)

constexpr uint64_t kKeyInf = ~0ull;
__device__ __forceinline__ uint32_t f32_to_ordered(float f) {
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ordered_to_f32(uint32_t o) {
  return __uint_as_float((o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o);
}
__device__ __forceinline__ uint64_t umin64(uint64_t a, uint64_t b) { return a < b ? a : b; }
__device__ __forceinline__ uint64_t umax64(uint64_t a, uint64_t b) { return a < b ? b : a; }

// ---- the shipped kernel (k_select.hip), one workgroup of four waves per query -------------------------------------
__global__ __launch_bounds__(256) void shipped_kernel(const float* __restrict__ scores, uint32_t n_rows, uint32_t q_rows,
                                                      uint32_t rank, float* __restrict__ thr) {
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

// ---- the candidate: one workgroup per 64 consecutive queries, lane = query ---------------------------------------------
// LDS: list[w][r][lane] (r = position in the wave's sorted list, lane fastest: conflict-free), cnt[w][lane]
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void transposed_kernel(const float* __restrict__ scores, uint32_t n_rows,
                                                               uint32_t q_rows, uint32_t nq, uint32_t rank,
                                                               float* __restrict__ thr) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const uint32_t q = blockIdx.x * 64u + (uint32_t)lane;
  float* L = lds + (size_t)w * 64 * 64 + lane;             // L[r * 64] = entry r of this lane's list
  uint32_t* cnts = (uint32_t*)(lds + (size_t)WAVES * 64 * 64);
  uint32_t cnt = 0;
  float worst = __builtin_inff();                            // the list's rank-th entry once it is full
  const float* col = scores + q;                             // (q < q_rows always: q_rows is a multiple of 64)
  constexpr int kAhead = 8;                                  // rows in flight per wave
  for (uint32_t r0 = (uint32_t)w * kAhead; r0 < n_rows; r0 += WAVES * kAhead) {
    float v[kAhead];
#pragma unroll
    for (int j = 0; j < kAhead; ++j) {
      const uint32_t row = r0 + (uint32_t)j;
      v[j] = row < n_rows ? col[(size_t)row * q_rows] : __builtin_inff();
    }
#pragma unroll
    for (int j = 0; j < kAhead; ++j) {
      const float sc = v[j];
      if (sc == sc && sc < worst) {                          // (NaN fails the first test, +inf the second)
        uint32_t i = cnt < rank ? cnt : rank - 1;            // a full list drops its last entry
        while (i > 0 && L[(size_t)(i - 1) * 64] > sc) {
          L[(size_t)i * 64] = L[(size_t)(i - 1) * 64];
          --i;
        }
        L[(size_t)i * 64] = sc;
        if (cnt < rank) ++cnt;
        if (cnt == rank) worst = L[(size_t)(rank - 1) * 64];
      }
    }
  }
  cnts[w * 64 + lane] = cnt;
  __syncthreads();
  if (w == 0 && q < nq) {
    // the rank-th smallest of the union of the WAVES sorted lists of this query
    uint32_t p[WAVES], c[WAVES];
#pragma unroll
    for (int x = 0; x < WAVES; ++x) {
      p[x] = 0;
      c[x] = cnts[x * 64 + lane];
    }
    float kth = __builtin_inff();
    for (uint32_t t = 0; t < rank; ++t) {
      float m = __builtin_inff();
      int from = -1;
#pragma unroll
      for (int x = 0; x < WAVES; ++x) {
        if (p[x] < c[x]) {
          const float h = lds[((size_t)x * 64 + p[x]) * 64 + lane];
          if (from < 0 || h < m) {
            m = h;
            from = x;
          }
        }
      }
      if (from < 0) {
        kth = __builtin_inff();                              // fewer than `rank` valid scores
        break;
      }
#pragma unroll
      for (int x = 0; x < WAVES; ++x)
        if (x == from) ++p[x];
      kth = m;
    }
    thr[q] = kth;
  }
}

static float host_kth(const std::vector<float>& s, uint32_t n_rows, uint32_t q_rows, uint32_t q, uint32_t rank) {
  std::vector<float> v;
  for (uint32_t r = 0; r < n_rows; ++r) {
    const float x = s[(size_t)r * q_rows + q];
    if (x == x && x < INFINITY) v.push_back(x);
  }
  if (v.size() < rank) return INFINITY;
  std::nth_element(v.begin(), v.begin() + (rank - 1), v.end());
  return v[rank - 1];
}

int main() {
  const uint32_t n_rows = 2048, q_rows = 1024, nq = 1000;
  std::vector<float> h((size_t)n_rows * q_rows);
  srand(7);
  for (auto& x : h) {
    float u = 0.0f;
    for (int i = 0; i < 6; ++i) u += (float)rand() / (float)RAND_MAX;
    x = (u - 3.0f) * 0.05f;                                 // lower-bound scores: bell-shaped around 0
  }
  for (uint32_t r = 2000; r < n_rows; ++r)                  // padding rows of a short sample: +inf
    for (uint32_t q = 0; q < q_rows; ++q) h[(size_t)r * q_rows + q] = INFINITY;
  for (uint32_t r = 0; r < n_rows; r += 97) h[(size_t)r * q_rows + 5] = NAN;              // a query with NaN scores
  for (uint32_t r = 3; r < n_rows; ++r) h[(size_t)r * q_rows + 9] = INFINITY;             // a query with 3 valid scores
  for (uint32_t r = 0; r < n_rows; ++r) h[(size_t)r * q_rows + 11] = 0.25f;               // all equal
  float *d_s, *d_a, *d_b;
  CHECK(hipMalloc((void**)&d_s, h.size() * 4));
  CHECK(hipMalloc((void**)&d_a, q_rows * 4));
  CHECK(hipMalloc((void**)&d_b, q_rows * 4));
  CHECK(hipMemcpy(d_s, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  constexpr int W = 4;
  const size_t lds = (size_t)W * 64 * 64 * 4 + (size_t)W * 64 * 4;
  CHECK(hipFuncSetAttribute((const void*)transposed_kernel<W>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  for (uint32_t rank : {8u, 16u, 64u}) {
    CHECK(hipMemset(d_a, 0xFF, q_rows * 4));
    CHECK(hipMemset(d_b, 0xFF, q_rows * 4));
    float ms_a = 0, ms_b = 0;
    for (int rep = 0; rep < 2; ++rep) {  // rep 0 warms up
      CHECK(hipEventRecord(e0));
      for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(shipped_kernel, dim3(nq), dim3(256), 0, 0, d_s, n_rows, q_rows, rank, d_a);
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      CHECK(hipEventElapsedTime(&ms_a, e0, e1));
      CHECK(hipEventRecord(e0));
      for (int i = 0; i < 200; ++i)
        hipLaunchKernelGGL(transposed_kernel<W>, dim3(q_rows / 64), dim3(64 * W), lds, 0, d_s, n_rows, q_rows, nq, rank, d_b);
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      CHECK(hipEventElapsedTime(&ms_b, e0, e1));
    }
    CHECK(hipGetLastError());
    std::vector<float> a(q_rows), b(q_rows);
    CHECK(hipMemcpy(a.data(), d_a, q_rows * 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(b.data(), d_b, q_rows * 4, hipMemcpyDeviceToHost));
    uint32_t bad_a = 0, bad_b = 0;
    for (uint32_t q = 0; q < nq; ++q) {
      const float want = host_kth(h, n_rows, q_rows, q, rank);
      bad_a += !(a[q] == want);
      bad_b += !(b[q] == want);
    }
    printf("rank %2u: shipped %.2f us per launch (%u of %u queries wrong), transposed %.2f us (%u wrong)\n", rank,
           ms_a * 5.0f, bad_a, nq, ms_b * 5.0f, bad_b);
  }
  return 0;
}
