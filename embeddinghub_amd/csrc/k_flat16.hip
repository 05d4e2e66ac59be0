// flat_scan16_kernel — the exhaustive scan as an fp16-MFMA FILTER (v_mfma_f32_32x32x16_f16, 16x the
// fp32 matrix rate) in front of the canonical fp32 re-rank.  Results stay those of the exact fp32
// scan: the filter only decides which rows become candidates, and it does so with a LOWER BOUND of
// every row's score, so a true top-k row can never be filtered out without the re-rank noticing
// (rerank16_kernel certifies each query; uncertified queries are re-run through the fp32 scan).
//
// The scan copy (k_misc.hip: make_scan16): every row normalised to unit length and rounded to
// binary16, X16[cap][ld16]; queries likewise, Q16[q_rows][ld16].  With n_r = |x_r|, beta = |q| and
// dot^ = <q^,x^> every metric of the engine is an affine function of dot^:
//     S(r,q) = b_r*gamma_q + a_r*dot^        D(r,q) = u_q*S + v_q   (u_q > 0)
//     cosine  a=-1    b=1     gamma=1          u=1      v=0
//     IP      a=-n_r  b=1     gamma=1/beta     u=beta   v=0
//     L2^2    a=-n_r  b=n_r^2 gamma=1/(2 beta) u=2 beta v=beta^2
// |dot16 - dot^| <= eps (fp16 rounding of two unit vectors: 2^-10 worst case by Cauchy-Schwarz, plus
// the fp32 norm / accumulation terms, ScanArgs16::eps), and a_r <= 0, so with the accumulators
// started at +eps the kernel's score  S_lower = b*gamma + a*(dot16 + eps)  never exceeds the true S.
//
// Kernel shape: the 8-wave mapping of k_flat8.hip with the same LDS image — a stage row is 128 bytes,
// here 64 halves (BK = 64): workgroup tile 128 rows x 256 queries, wave tile 64 x 64 = 2x2 MFMA
// blocks, 3-deep ring of 48-KiB stages filled by global->LDS DMA, one raw s_barrier per stage.  A
// stage is 16 MFMAs (4 k-steps x 4 blocks) instead of 64, so the fragment reads of k-step j+1 and
// the DMA duty are spread over the 4 MFMAs of k-step j.
// Epilogue: phase 1 reduces the wave's 64x64 scores to one min per (lane, query column) and tests it
// against the list threshold — no per-score bookkeeping; only when some lane hits (rare once the
// sample pass has set the thresholds) the per-score bit masks are built and the hits appended.
#include "ehx_kernels.h"
#include "k_scan_common.h"

namespace ehx {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int kThreads16 = 512;
constexpr uint32_t kRing16 = 3;
constexpr uint32_t kBK16 = 64;                                       // halves per stage row (128 B)
constexpr uint32_t kXStage16 = kTileRows * 128;                      // 16 KiB
constexpr uint32_t kQStage16 = kTileQ * 128;                         // 32 KiB
constexpr uint32_t kXOff16 = 0;
constexpr uint32_t kQOff16 = kRing16 * kXStage16;
constexpr uint32_t kThrKeyOff16 = kQOff16 + kRing16 * kQStage16;     // u64 thr_key[512]
constexpr uint32_t kThrFOff16 = kThrKeyOff16 + kLists8 * 8;          // f32 thr_f[512]
constexpr uint32_t kCntOff16 = kThrFOff16 + kLists8 * 4;             // i32 cnt[512]
constexpr uint32_t kFlagOff16 = kCntOff16 + kLists8 * 4;             // i32 simd_rank[4] (+pad)
constexpr uint32_t kRowpOff16 = kFlagOff16 + 32;                     // float2 rowp_lds[4][128]
constexpr uint32_t kLdsBytes16 = kRowpOff16 + 4 * 128 * 8;
static_assert(kLdsBytes16 <= 160 * 1024, "LDS budget");

__device__ __forceinline__ f16x8 frag16(const char* p) { return *(const f16x8*)p; }

#define EHX_MFMA16(A, B, C) __builtin_amdgcn_mfma_f32_32x32x16_f16((A), (B), (C), 0, 0, 0)

}  // namespace

size_t scan16_lds_bytes() { return kLdsBytes16; }

template <bool COS>
__global__ __launch_bounds__(kThreads16, 2) void flat_scan16_kernel(const ScanArgs16 a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = w >> 2, wc = w & 3;
  const int h = lane >> 5, i31 = lane & 31;

  uint32_t qt, chunk;
  {
    const uint32_t b = blockIdx.x;
    if (a.xcd_map) {
      const uint32_t xcd = b & 7u, slot = b >> 3;
      qt = slot % a.q_tiles;
      chunk = xcd * (a.n_chunks >> 3) + slot / a.q_tiles;
    } else {
      qt = b % a.q_tiles;
      chunk = b / a.q_tiles;
    }
  }
  uint64_t* thr_key = (uint64_t*)(smem + kThrKeyOff16);
  float* thr_f = (float*)(smem + kThrFOff16);
  int* cnt = (int*)(smem + kCntOff16);
  int* simd_rank = (int*)(smem + kFlagOff16);
  const float2* rowp_lds = (const float2*)(smem + kRowpOff16);
  uint64_t* cand = a.cand + (size_t)blockIdx.x * ((size_t)kLists8 * kCandSlots);
  unsigned long long* gthr = a.gthr + (size_t)qt * kTileQ;

  {  // 512 threads, 512 lists; start from the query's global threshold (the sample pass set it)
    const int wq = ((tid >> 6) & 3) * 64 + (tid & 63);
    const unsigned long long g = gthr[wq];
    thr_key[tid] = g;
    thr_f[tid] = g == kKeyInf ? __builtin_inff() : ordered_to_f32((uint32_t)(g >> 32));
    cnt[tid] = 0;
  }
  if (tid < 4) simd_rank[tid] = 0;
  __syncthreads();
  // early/late DMA duty by actual SIMD co-residency, as in k_flat8.hip
  const int simd_id = (int)__builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4);
  int my_rank = 0;
  if (lane == 0) my_rank = atomicAdd(&simd_rank[simd_id & 3], 1);
  const bool late = (__builtin_amdgcn_readfirstlane(my_rank) & 1) != 0;

  // per-query gamma of this lane's two query columns (wc*64 + cb*32 + i31)
  const float gam0 = a.qgamma[(size_t)qt * kTileQ + wc * 64 + i31];
  const float gam1 = a.qgamma[(size_t)qt * kTileQ + wc * 64 + 32 + i31];

  const uint32_t tile_begin = a.tile0 + chunk * a.tiles_per_chunk;
  uint32_t tile_end = tile_begin + a.tiles_per_chunk;
  if (tile_end > a.tile0 + a.n_tiles) tile_end = a.tile0 + a.n_tiles;
  const uint32_t my_tiles = tile_end > tile_begin ? tile_end - tile_begin : 0u;
  const uint32_t ktiles = a.ld / kBK16;
  const uint32_t total_steps = my_tiles * ktiles;

  // ---- DMA duty of this wave: X pieces 2w, 2w+1; Q pieces 4w..4w+3; wave 0 also the row parameters.
  // piece `ins` = 8 stage rows x 128 B; lane L -> row 8*ins + (L>>3), physical 16-B chunk p = L&7 holds
  // logical chunk p ^ ((row>>1)&7): only the parity of ins matters -> two lane offsets.
  const uint32_t row_bytes = a.ld * 2u;
  const char* Qtile = (const char*)a.Q + (size_t)qt * kTileQ * row_bytes;
  const char* Xbase = (const char*)a.X + (size_t)tile_begin * kTileRows * row_bytes;
  const float2* Rbase = a.rowp + (size_t)tile_begin * kTileRows;
  const size_t tile_stride = (size_t)kTileRows * row_bytes;
  const uint32_t c0 = (uint32_t)(lane & 7) ^ (uint32_t)(lane >> 4);
  const uint32_t lane_row = (uint32_t)(lane >> 3) * row_bytes;
  const uint32_t l_even = lane_row + c0 * 16u;
  const uint32_t l_odd = lane_row + (c0 ^ 4u) * 16u;
  const uint32_t piece_stride = 8u * row_bytes;
  uint32_t pre_t = 0, pre_kt = 0, pre_buf = 0, issued = 0;

#define EHX_PIECE16(U)                                                                                  \
  do {                                                                                                  \
    if ((U) < 2) {                                                                                      \
      const char* Xt = Xbase + pre_t * tile_stride + (size_t)pre_kt * 128u;                             \
      glds16_8(Xt + (size_t)((2 * w + (U)) * piece_stride) + (((U) & 1) ? l_odd : l_even),              \
               smem + kXOff16 + pre_buf * kXStage16 + (2 * w + (U)) * 1024);                            \
    } else if ((U) < 6) {                                                                               \
      const char* Qt = Qtile + (size_t)pre_kt * 128u;                                                   \
      glds16_8(Qt + (size_t)((4 * w + (U) - 2) * piece_stride) + ((((U) - 2) & 1) ? l_odd : l_even),    \
               smem + kQOff16 + pre_buf * kQStage16 + (4 * w + (U) - 2) * 1024);                        \
    } else if (w == 0) {                                                                                \
      glds16_8((const char*)(Rbase + pre_t * kTileRows) + lane * 16,                                    \
               smem + kRowpOff16 + (pre_t & 3u) * 1024u);                                               \
    }                                                                                                   \
  } while (0)
#define EHX_STAGE_ADVANCE16()                            \
  do {                                                   \
    if (++pre_kt == ktiles) {                            \
      pre_kt = 0;                                        \
      ++pre_t;                                           \
    }                                                    \
    pre_buf = pre_buf == kRing16 - 1 ? 0u : pre_buf + 1; \
    ++issued;                                            \
  } while (0)

  // ---- fragment read constants: k-step j of a stage row is the 32-byte span of logical chunks 2j, 2j+1;
  // lanes 0-31 (h = 0) take chunk 2j, lanes 32-63 chunk 2j+1 (any fixed k permutation is fine as long
  // as rows and queries use the same one)
  const uint32_t hs = (uint32_t)h ^ ((uint32_t)(i31 >> 1) & 7u);
  uint32_t joff[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) joff[j] = (((uint32_t)(2 * j)) ^ hs) * 16;
  const uint32_t a_row_off = (uint32_t)(wr * 64 + i31) * 128;  // + rb*4096
  const uint32_t b_row_off = (uint32_t)(wc * 64 + i31) * 128;  // + cb*4096

  f32x16 acc[2][2];
  const float acc0 = a.eps;  // accumulators start at +eps: the scores are lower bounds
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[rb][cb][r] = acc0;

  // =============================== tile epilogue ===============================
  auto epilogue = [&](uint32_t t) {
    const uint32_t tile_row0 = (tile_begin + t) * kTileRows;
    const float2* rp = rowp_lds + (t & 3u) * 128u;
    const int lbase = w * 64 + i31;  // + cb*32
    const float thrf0 = thr_f[lbase], thrf1 = thr_f[lbase + 32];
    // ---- phase 1: one min per query column ----
    bool hit;
    if (COS) {
      // valid rows have (a, b) = (-1, 1) and gamma = 1: min S = 1 - max acc, no row parameters needed
      // (padding / invalid rows may raise a false alarm; phase 2 evaluates them properly)
      float m0 = acc[0][0][0], m1 = acc[0][1][0];
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          m0 = fmaxf(m0, acc[rb][0][reg]);
          m1 = fmaxf(m1, acc[rb][1][reg]);
        }
      }
      hit = (__builtin_fmaf(-1.0f, m0, 1.0f) <= thrf0) | (__builtin_fmaf(-1.0f, m1, 1.0f) <= thrf1);
    } else {
      float m0 = __builtin_inff(), m1 = __builtin_inff();
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const uint32_t r = (uint32_t)(wr * 64 + rb * 32 + (reg & 3) + 8 * (reg >> 2)) + 4u * h;
          const float2 ab = rp[r];
          m0 = fminf(m0, __builtin_fmaf(ab.x, acc[rb][0][reg], ab.y * gam0));
          m1 = fminf(m1, __builtin_fmaf(ab.x, acc[rb][1][reg], ab.y * gam1));
        }
      }
      hit = (m0 <= thrf0) | (m1 <= thrf1);
    }
    if (!__any(hit)) return;  // common case once the thresholds are warm
    // ---- slow path: per-score bit masks (word = rb, bit = cb*16 + reg), then the appends ----
    uint32_t pend[2] = {0u, 0u};
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const uint32_t r = (uint32_t)(wr * 64 + rb * 32 + (reg & 3) + 8 * (reg >> 2)) + 4u * h;
        const float2 ab = rp[r];
        const float s0 = __builtin_fmaf(ab.x, acc[rb][0][reg], ab.y * gam0);
        const float s1 = __builtin_fmaf(ab.x, acc[rb][1][reg], ab.y * gam1);
        pend[rb] |= (s0 <= thrf0) ? (1u << reg) : 0u;
        pend[rb] |= (s1 <= thrf1) ? (1u << (16 + reg)) : 0u;
      }
    }
    if (!__any((pend[0] | pend[1]) != 0u)) return;
    const int trigger = (int)a.kprime + ((int)kCandSlots - (int)a.kprime) / 2;
    for (int round = 0; round < 1024; ++round) {
      bool hit_trigger = false;
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
        uint32_t retry = 0u;
        while (__any(pend[rb] != 0u)) {
          if (pend[rb] != 0u) {
            const int b = __builtin_ctz(pend[rb]);
            pend[rb] &= pend[rb] - 1u;
            float dot = 0.0f;
#pragma unroll
            for (int i = 0; i < 16; ++i) dot = (b == i) ? acc[rb][0][i] : dot;
#pragma unroll
            for (int i = 0; i < 16; ++i) dot = (b == 16 + i) ? acc[rb][1][i] : dot;
            const int reg = b & 15;
            const int cb = b >> 4;
            const uint32_t r = (uint32_t)(wr * 64 + rb * 32 + (reg & 3) + 8 * (reg >> 2)) + 4u * h;
            const float2 ab = rp[r];
            const float sc = __builtin_fmaf(ab.x, dot, ab.y * (cb ? gam1 : gam0));
            const int pos = scan8_push(sc, tile_row0 + r, lbase + cb * 32, a.n, cand, cnt, thr_key);
            if (pos >= (int)kCandSlots) retry |= 1u << b;  // list full: compact, then try again
            hit_trigger |= pos + 1 >= trigger;
          }
        }
        pend[rb] = retry;
      }
      if (!__any(hit_trigger)) return;
      scan8_compact(w, wc, lane, (int)a.kprime, false, cand, cnt, thr_key, thr_f, gthr);
      if (!__any((pend[0] | pend[1]) != 0u)) return;
    }
    if (lane == 0) atomicAdd(a.err, 1u);  // never reached: a compacted list has free slots
  };

  constexpr int kP = 6;  // X + Q pieces of a wave per stage; wave 0 adds the row-parameter piece

  // ---- prologue: every wave issues its share of the first (up to) three stages ----
  while (issued < total_steps && issued < kRing16) {
#pragma unroll
    for (int u = 0; u < 7; ++u) EHX_PIECE16(u);
    EHX_STAGE_ADVANCE16();
  }
  if (w == 0) {
    if (issued >= 3) wait_vmcnt<2 * (kP + 1)>();
    else if (issued == 2) wait_vmcnt<kP + 1>();
    else wait_vmcnt<0>();
  } else {
    if (issued >= 3) wait_vmcnt<2 * kP>();
    else if (issued == 2) wait_vmcnt<kP>();
    else wait_vmcnt<0>();
  }
  hot_barrier();  // B_0 (also publishes the state init)

  // fragment sets: F0 / F1 alternate between k-steps
  f16x8 fa0[2], fb0[2], fa1[2], fb1[2];
  if (total_steps > 0) {
    fa0[0] = frag16(smem + kXOff16 + a_row_off + joff[0]);
    fa0[1] = frag16(smem + kXOff16 + a_row_off + 4096 + joff[0]);
    fb0[0] = frag16(smem + kQOff16 + b_row_off + joff[0]);
    fb0[1] = frag16(smem + kQOff16 + b_row_off + 4096 + joff[0]);
  }

  // MFMA number m (0..3) of a k-step on fragment set (A,B): block (rb, cb) = (m&1, m>>1)
#define EHX_ONE16(A, B, M) acc[(M) & 1][(M) >> 1] = EHX_MFMA16(A[(M) & 1], B[(M) >> 1], acc[(M) & 1][(M) >> 1])
#define EHX_KSTEP16(A, B, An, Bn, XS, QS)        \
  do {                                           \
    An[0] = frag16((XS));                        \
    An[1] = frag16((XS) + 4096);                 \
    Bn[0] = frag16((QS));                        \
    Bn[1] = frag16((QS) + 4096);                 \
    EHX_ONE16(A, B, 0);                          \
    EHX_ONE16(A, B, 1);                          \
    EHX_ONE16(A, B, 2);                          \
    EHX_ONE16(A, B, 3);                          \
    __builtin_amdgcn_sched_barrier(0);           \
  } while (0)

  uint32_t kt = 0, t = 0, buf = 0;
  for (uint32_t step = 0; step < total_steps; ++step) {
    const uint32_t nbuf = buf == kRing16 - 1 ? 0u : buf + 1;
    const char* xs = smem + kXOff16 + buf * kXStage16 + a_row_off;
    const char* qs = smem + kQOff16 + buf * kQStage16 + b_row_off;
    const char* xn = smem + kXOff16 + nbuf * kXStage16 + a_row_off;
    const char* qn = smem + kQOff16 + nbuf * kQStage16 + b_row_off;
    const bool has_next = step + 1 < total_steps;
    // every MFMA is emitted exactly once; only DMA pieces and the next stage's reads sit behind
    // wave-uniform flags (see k_flat8.hip)
    const bool late_dma = late && step >= 1 && issued < total_steps;
    const bool early_dma = !late && has_next && issued < total_steps;

    // ---- k-step 0: the late waves do their DMA duty here ----
    fa1[0] = frag16(xs + joff[1]);
    fa1[1] = frag16(xs + 4096 + joff[1]);
    fb1[0] = frag16(qs + joff[1]);
    fb1[1] = frag16(qs + 4096 + joff[1]);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      EHX_ONE16(fa0, fb0, m);
      if (late_dma) {
        EHX_PIECE16(2 * m);
        if (2 * m + 1 < 7) EHX_PIECE16(2 * m + 1);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (late_dma) EHX_STAGE_ADVANCE16();
    EHX_KSTEP16(fa1, fb1, fa0, fb0, xs + joff[2], qs + joff[2]);
    EHX_KSTEP16(fa0, fb0, fa1, fb1, xs + joff[3], qs + joff[3]);

    // ---- stage barrier, then k-step 3 with the next stage's first fragments and (early waves) the
    // DMA pieces of stage step+3 ----
    if (has_next) {
      if (step + 2 < total_steps) {
        if (w == 0) wait_vmcnt<kP + 1>();
        else wait_vmcnt<kP>();
      } else {
        wait_vmcnt<0>();
      }
      hot_barrier();  // B_{step+1}: stage step+1 visible; ring slot `buf` is free again
      fa0[0] = frag16(xn + joff[0]);
      fa0[1] = frag16(xn + 4096 + joff[0]);
      fb0[0] = frag16(qn + joff[0]);
      fb0[1] = frag16(qn + 4096 + joff[0]);
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      EHX_ONE16(fa1, fb1, m);
      if (early_dma) {
        EHX_PIECE16(2 * m);
        if (2 * m + 1 < 7) EHX_PIECE16(2 * m + 1);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (early_dma) EHX_STAGE_ADVANCE16();
    buf = nbuf;
    if (++kt == ktiles) {
      kt = 0;
      epilogue(t);
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[rb][cb][r] = acc0;
      ++t;
    }
  }
#undef EHX_KSTEP16
#undef EHX_ONE16
#undef EHX_PIECE16
#undef EHX_STAGE_ADVANCE16

  // ---- final: sort this wave's 64 lists and publish them: part[q][chunk*2 + wr][k'] ----
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  for (int ql = 0; ql < 64; ++ql) {
    const int list = w * 64 + ql;
    const int cq = cnt[list];
    const int nv = cq < (int)kCandSlots ? cq : (int)kCandSlots;
    uint64_t key = lane < nv ? cand[list * kCandSlots + lane] : kKeyInf;
    key = wave_sort64_8(key, lane);
    if (lane < (int)a.kprime)
      a.part[((size_t)(qt * kTileQ + wc * 64 + ql) * a.lists_total + a.list0 + chunk * 2 + wr) * a.kprime + lane] = key;
  }
}

hipError_t launch_flat_scan16(const ScanArgs16& a, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)flat_scan16_kernel<false>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes16);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void*)flat_scan16_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)kLdsBytes16);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  const uint32_t grid = a.q_tiles * a.n_chunks;
  if (a.cos) hipLaunchKernelGGL(flat_scan16_kernel<true>, dim3(grid), dim3(kThreads16), kLdsBytes16, st, a);
  else hipLaunchKernelGGL(flat_scan16_kernel<false>, dim3(grid), dim3(kThreads16), kLdsBytes16, st, a);
  return hipGetLastError();
}

}  // namespace ehx
