// flat_scan8_kernel — the exhaustive scan of k_flat.hip re-mapped to 8 waves (two per SIMD).
//
// Why a second mapping: rocprofv3 on the 4-wave kernel (one wave per SIMD, profiles/r01_a) showed the
// matrix pipe busy only 80 % of the time; the rest is the compute waves' own global->LDS DMA issue
// (a 1-KiB piece blocks its issuing wave for 60-180 cycles, 12 pieces per wave per stage) plus the
// per-stage barrier and the tile epilogue — with a single wave per SIMD nothing else can issue MFMAs
// meanwhile.  Here every SIMD hosts two waves whose DMA duty is staggered half a stage apart
// (waves 0-3 issue right after the stage barrier, waves 4-7 one group later), so one wave's DMA issue
// always runs beside its sibling's MFMA issue.
//
//   * workgroup = 512 threads = 8 waves, 1 workgroup per CU; wave (wr, wc) = (w>>2, w&3) owns
//     64 rows x 64 queries = 2x2 MFMA 32x32 blocks (64 accumulator registers); workgroup tile
//     128 rows x 256 queries, BK = 32 (same tile, same LDS image, same swizzle as k_flat.hip);
//   * LDS: 3-deep ring of stages (3 x 48 KiB) so a stage has two full stages to land;
//   * every wave DMA-copies its share of a stage (2 X pieces + 4 Q pieces; wave 0 also the tile's
//     row-parameter piece) and accounts for it with a counted s_waitcnt vmcnt before the single
//     workgroup barrier per stage (raw s_barrier: __syncthreads() would drain the DMA queue);
//   * fragments of group g+1 are read from LDS while group g's 16 MFMAs issue.
// Epilogue, candidate slots, key packing, output format: identical to flat_scan_kernel.
#include "ehx_kernels.h"
#include "k_scan_common.h"

namespace ehx {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int kThreads8 = 512;
constexpr uint32_t kRing8 = 3;
constexpr uint32_t kXStage8 = kTileRows * kBK * 4;                 // 16 KiB
constexpr uint32_t kQStage8 = kTileQ * kBK * 4;                    // 32 KiB
constexpr uint32_t kXOff8 = 0;
constexpr uint32_t kQOff8 = kRing8 * kXStage8;
constexpr uint32_t kThrKeyOff8 = kQOff8 + kRing8 * kQStage8;       // u64 thr_key[512]
constexpr uint32_t kThrFOff8 = kThrKeyOff8 + kLists8 * 8;          // f32 thr_f[512]
constexpr uint32_t kCntOff8 = kThrFOff8 + kLists8 * 4;             // i32 cnt[512]
constexpr uint32_t kFlagOff8 = kCntOff8 + kLists8 * 4;             // i32 flags[4]
constexpr uint32_t kRowpOff8 = kFlagOff8 + 32;                     // (flags[4], simd_rank[4]) then float2 rowp_lds[4][128]
constexpr uint32_t kLdsBytes8 = kRowpOff8 + 4 * 128 * 8;
static_assert(kLdsBytes8 <= 160 * 1024, "LDS budget");

__device__ __forceinline__ f32x4 frag_read(const char* p) {
#if (EHX_ABL & 4)
  f32x4 v = {1.0f, 2.0f, 3.0f, 4.0f};
  asm volatile("" : "+v"(v));
  return v;
#else
  return *(const f32x4*)p;
#endif
}

// fp16 rows: 4 consecutive halves (8 bytes) widened exactly to fp32
__device__ __forceinline__ f32x4 frag_read_h(const char* p) {
#if (EHX_ABL & 4)
  return frag_read(p);
#else
  const uint2 u = *(const uint2*)p;
  const __half2 lo = *(const __half2*)&u.x, hi = *(const __half2*)&u.y;
  const float2 a = __half22float2(lo), b = __half22float2(hi);
  return (f32x4){a.x, a.y, b.x, b.y};
#endif
}

#define EHX_MFMA8(A, B, C) __builtin_amdgcn_mfma_f32_32x32x2f32((A), (B), (C), 0, 0, 0)
// Ablation hooks for profiling builds only (scripts/ablate_scan.sh); the shipped library defines none.
#ifndef EHX_ABL
#define EHX_ABL 0
#endif
#define ABL_NO_EPILOGUE (EHX_ABL & 1)
#define ABL_NO_DMA (EHX_ABL & 2)
#define ABL_NO_LDSREAD (EHX_ABL & 4)
#define ABL_NO_BARRIER (EHX_ABL & 8)
#define ABL_EPI_PHASE1_ONLY (EHX_ABL & 16)
#define ABL_EPI_NO_PHASE1 (EHX_ABL & 32)
#define ABL_COUNT (EHX_ABL & 64)  // instrumentation: a.gthr[q_rows + 0..3] = hot tiles, push iterations, pushes, compactions

}  // namespace

size_t scan8_lds_bytes() { return kLdsBytes8; }

template <bool HALF>
__global__ __launch_bounds__(kThreads8, 2) void flat_scan8_kernel(const ScanArgs a) {
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
  uint64_t* thr_key = (uint64_t*)(smem + kThrKeyOff8);
  float* thr_f = (float*)(smem + kThrFOff8);
  int* cnt = (int*)(smem + kCntOff8);
  int* flags = (int*)(smem + kFlagOff8);
  int* simd_rank = flags + 4;
  const float2* rowp_lds = (const float2*)(smem + kRowpOff8);
  uint64_t* cand = a.cand + (size_t)blockIdx.x * ((size_t)kLists8 * kCandSlots);
  unsigned long long* gthr = a.gthr + (size_t)qt * kTileQ;  // global per-query thresholds of this query tile

  {  // 512 threads, 512 lists; start from the query's global threshold (the sample pass set it)
    const int wq = ((tid >> 6) & 3) * 64 + (tid & 63);  // list tid belongs to wave tid>>6, query wc*64 + ql
    const unsigned long long g = gthr[wq];
    thr_key[tid] = g;
    thr_f[tid] = g == kKeyInf ? __builtin_inff() : ordered_to_f32((uint32_t)(g >> 32));
    cnt[tid] = 0;
  }
  if (tid < 4) flags[tid] = 0;
  if (tid < 4) simd_rank[tid] = 0;
  __syncthreads();
  // Which of the two waves sharing this wave's SIMD am I?  (HW_REG_HW_ID.SIMD_ID, bits 5:4.)  The two
  // do their DMA duty half a stage apart so that one's DMA issue overlaps the other's MFMA issue.
  const int simd_id = (int)__builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4);
  int my_rank = 0;
  if (lane == 0) my_rank = atomicAdd(&simd_rank[simd_id & 3], 1);
  const bool late = (__builtin_amdgcn_readfirstlane(my_rank) & 1) != 0;

  const uint32_t tile_begin = a.tile0 + chunk * a.tiles_per_chunk;
  uint32_t tile_end = tile_begin + a.tiles_per_chunk;
  if (tile_end > a.tile0 + a.n_tiles) tile_end = a.tile0 + a.n_tiles;
  const uint32_t my_tiles = tile_end > tile_begin ? tile_end - tile_begin : 0u;
  const uint32_t ktiles = a.ld / kBK;
  const uint32_t total_steps = my_tiles * ktiles;

  // ---- DMA duty of this wave: X pieces 2w, 2w+1; Q pieces 4w..4w+3; wave 0 also the row parameters.
  // piece `ins` = 8 tile rows x 128 B; lane L -> row 8*ins + (L>>3), physical chunk p = L&7, logical
  // chunk c = p ^ ((4*ins + (L>>4)) & 7): only the parity of ins matters -> two lane offsets.
  const float* Qtile = a.Q + (size_t)qt * kTileQ * a.ld;
  // X rows: fp32 (128-B stage rows, 8 rows per DMA piece, 16 pieces per stage, 2 per wave) or fp16 (64-B
  // stage rows, 16 rows per piece, 8 pieces per stage, 1 per wave; 16-B chunk p of a row holds logical
  // chunk p ^ ((row>>2)&3))
  constexpr uint32_t kXElem = HALF ? 2u : 4u;
  const char* Xbase = (const char*)a.X + (size_t)tile_begin * kTileRows * a.ld * kXElem;
  const float2* Rbase = a.rowp + (size_t)tile_begin * kTileRows;
  const size_t tile_stride = (size_t)kTileRows * a.ld * kXElem;  // bytes
  const uint32_t lh = ((uint32_t)(lane >> 2) * a.ld * 2u) + (((uint32_t)(lane & 3) ^ ((uint32_t)(lane >> 4) & 3u)) * 16u);  // fp16 piece: lane byte offset
  const uint32_t piece_stride_h = 16u * a.ld * 2u;
  const uint32_t c0 = (uint32_t)(lane & 7) ^ (uint32_t)(lane >> 4);
  const uint32_t lane_row = (uint32_t)(lane >> 3) * a.ld;
  const uint32_t l_even = (lane_row + c0 * 4u) * 4u;  // bytes
  const uint32_t l_odd = (lane_row + (c0 ^ 4u) * 4u) * 4u;
  const uint32_t piece_stride = 8u * a.ld * 4u;
  uint32_t pre_t = 0, pre_kt = 0, pre_buf = 0, issued = 0;

  // piece u of this wave's duty for the stage (pre_t, pre_kt) -> ring slot pre_buf; u in 0..5 (+6)
#define EHX_PIECE(U)                                                                                    \
  do {                                                                                                  \
    if (ABL_NO_DMA) break;                                                                              \
    if ((U) < 2) {                                                                                      \
      const char* Xt = Xbase + pre_t * tile_stride + (size_t)pre_kt * kBK * kXElem;                     \
      if (!HALF) {                                                                                      \
        glds16_8(Xt + (size_t)((2 * w + (U)) * piece_stride) + (((U) & 1) ? l_odd : l_even),            \
                 smem + kXOff8 + pre_buf * kXStage8 + (2 * w + (U)) * 1024);                            \
      } else if ((U) == 0) {                                                                            \
        glds16_8(Xt + (size_t)(w * piece_stride_h) + lh, smem + kXOff8 + pre_buf * kXStage8 + w * 1024); \
      }                                                                                                 \
    } else if ((U) < 6) {                                                                               \
      const char* Qt = (const char*)(Qtile + pre_kt * kBK);                                             \
      glds16_8(Qt + (size_t)((4 * w + (U) - 2) * piece_stride) + ((((U) - 2) & 1) ? l_odd : l_even),    \
               smem + kQOff8 + pre_buf * kQStage8 + (4 * w + (U) - 2) * 1024);                          \
    } else if (w == 0) {                                                                                \
      glds16_8((const char*)(Rbase + pre_t * kTileRows) + lane * 16,                                    \
               smem + kRowpOff8 + (pre_t & 3u) * 1024u);                                                \
    }                                                                                                   \
  } while (0)
#define EHX_STAGE_ADVANCE()                         \
  do {                                              \
    if (++pre_kt == ktiles) {                       \
      pre_kt = 0;                                   \
      ++pre_t;                                      \
    }                                               \
    pre_buf = pre_buf == kRing8 - 1 ? 0u : pre_buf + 1; \
    ++issued;                                       \
  } while (0)

  // pieces this wave has in flight per stage (for the counted vmcnt waits)
  constexpr int kP = HALF ? 5 : 6;  // X + Q pieces of a wave; wave 0 adds the row-parameter piece

  // ---- fragment read constants ----
  const uint32_t hs = (uint32_t)h ^ ((uint32_t)(i31 >> 1) & 7u);
  uint32_t joff[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) joff[j] = (((uint32_t)(2 * j)) ^ hs) * 16;
  // A (row) fragments: fp32 stage rows are 128 B (b128 read at the swizzled 16-B chunk), fp16 stage rows
  // are 64 B (b64 read: chunk j ^ ((row>>2)&3), 8-byte half h)
  const uint32_t a_row_off = (uint32_t)(wr * 64 + i31) * (HALF ? 64u : 128u);  // + rb*kXrb
  constexpr uint32_t kXrb = HALF ? 2048u : 4096u;
  uint32_t joffa[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
    joffa[j] = HALF ? ((((uint32_t)j) ^ ((uint32_t)(i31 >> 2) & 3u)) * 16u + (uint32_t)h * 8u) : joff[j];
#define EHX_XREAD(P) (HALF ? frag_read_h((const char*)(P)) : frag_read((const char*)(P)))
  const uint32_t b_row_off = (uint32_t)(wc * 64 + i31) * 128;  // + cb*4096

  f32x16 acc[2][2];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[rb][cb][r] = 0.0f;

  // =============================== tile epilogue ===============================
  // Entirely wave-local (no workgroup barrier): every wave owns the candidate lists of ITS 64 queries
  // for ITS 64 rows of the tile.
  // Phase 1 (branch-free, unrolled): approximate distance s = dot*a_row + b_row of every accumulator,
  // recorded as one bit "s <= threshold of its list" (word = rb, bit = cb*16 + reg).
  // Phase 2 (rare): lanes with set bits extract the dot product with a select chain, recompute s
  // with the same fma and append the (score,id) key to the list (LDS atomic slot + 8-byte store);
  // a list that reaches the trigger is compacted on the spot by the wave.
  auto epilogue = [&](uint32_t t) {
    const uint32_t tile_row0 = (tile_begin + t) * kTileRows;
    const float2* rp = rowp_lds + (t & 3u) * 128u;
    const int lbase = w * 64 + i31;  // + cb*32
    uint32_t pend[2] = {0u, 0u};  // word = rb, bit = cb*16 + reg
    if (!ABL_EPI_NO_PHASE1) {
      const float thrf0 = thr_f[lbase], thrf1 = thr_f[lbase + 32];
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const uint32_t r = (uint32_t)(wr * 64 + rb * 32 + (reg & 3) + 8 * (reg >> 2)) + 4u * h;
          const float2 ab = rp[r];
          const float s0 = __builtin_fmaf(acc[rb][0][reg], ab.x, ab.y);
          const float s1 = __builtin_fmaf(acc[rb][1][reg], ab.x, ab.y);
          pend[rb] |= (s0 <= thrf0) ? (1u << reg) : 0u;
          pend[rb] |= (s1 <= thrf1) ? (1u << (16 + reg)) : 0u;
        }
      }
    }
    if (ABL_EPI_PHASE1_ONLY) {
      asm volatile("" ::"v"(pend[0]), "v"(pend[1]));
      return;
    }
    if (!__any((pend[0] | pend[1]) != 0u)) return;  // common case once the thresholds are warm
    if (ABL_COUNT && lane == 0) atomicAdd(&a.gthr[a.q_tiles * 256 + 0], 1ull);
    const int trigger = (int)a.kprime + ((int)kCandSlots - (int)a.kprime) / 2;
    for (int round = 0; round < 1024; ++round) {
      bool hit_trigger = false;
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
        uint32_t retry = 0u;
        while (__any(pend[rb] != 0u)) {
          if (ABL_COUNT && lane == 0) atomicAdd(&a.gthr[a.q_tiles * 256 + 1], 1ull);
          if (pend[rb] != 0u) {
            if (ABL_COUNT) atomicAdd(&a.gthr[a.q_tiles * 256 + 2], 1ull);
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
            const float sc = __builtin_fmaf(dot, ab.x, ab.y);
            const int pos = scan8_push(sc, tile_row0 + r, lbase + cb * 32, a.n, cand, cnt, thr_key);
            if (pos >= (int)kCandSlots) retry |= 1u << b;  // list full: compact, then try again
            hit_trigger |= pos + 1 >= trigger;
          }
        }
        pend[rb] = retry;
      }
      if (!__any(hit_trigger)) return;
      if (ABL_COUNT && lane == 0) atomicAdd(&a.gthr[a.q_tiles * 256 + 3], 1ull);
      scan8_compact(w, wc, lane, (int)a.kprime, false, cand, cnt, thr_key, thr_f, gthr);
      if (!__any((pend[0] | pend[1]) != 0u)) return;
    }
    if (lane == 0) atomicAdd(a.err, 1u);  // never reached: a compacted list has free slots
  };

  // ---- prologue: every wave issues its share of the first (up to) three stages ----
  while (issued < total_steps && issued < kRing8) {
#pragma unroll
    for (int u = 0; u < 7; ++u) EHX_PIECE(u);
    EHX_STAGE_ADVANCE();
  }
  // stage 0 landed <=> at most the pieces of the younger stages are still in flight
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

  f32x4 fa0[2], fb0[2], fa1[2], fb1[2];
  if (total_steps > 0) {
    fa0[0] = EHX_XREAD(smem + kXOff8 + a_row_off + joffa[0]);
    fa0[1] = EHX_XREAD(smem + kXOff8 + a_row_off + kXrb + joffa[0]);
    fb0[0] = frag_read((const char*)(smem + kQOff8 + b_row_off + joff[0]));
    fb0[1] = frag_read((const char*)(smem + kQOff8 + b_row_off + 4096 + joff[0]));
  }

  // one group: 16 MFMAs on (A,B); the 4 fragment reads of the next group in their shadow
#define EHX_GROUP8(A, B, An, Bn, XS, QS)                                  \
  do {                                                                    \
    An[0] = EHX_XREAD((XS));                                                       \
    An[1] = EHX_XREAD((XS) + kXrb);                                                \
    _Pragma("unroll") for (int tt = 0; tt < 2; ++tt) {                   \
      acc[0][0] = EHX_MFMA8(A[0][tt], B[0][tt], acc[0][0]);               \
      acc[1][0] = EHX_MFMA8(A[1][tt], B[0][tt], acc[1][0]);               \
      acc[0][1] = EHX_MFMA8(A[0][tt], B[1][tt], acc[0][1]);               \
      acc[1][1] = EHX_MFMA8(A[1][tt], B[1][tt], acc[1][1]);               \
    }                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                    \
    Bn[0] = frag_read((const char*)((QS)));                                        \
    Bn[1] = frag_read((const char*)((QS) + 4096));                                 \
    _Pragma("unroll") for (int tt = 2; tt < 4; ++tt) {                   \
      acc[0][0] = EHX_MFMA8(A[0][tt], B[0][tt], acc[0][0]);               \
      acc[1][0] = EHX_MFMA8(A[1][tt], B[0][tt], acc[1][0]);               \
      acc[0][1] = EHX_MFMA8(A[0][tt], B[1][tt], acc[0][1]);               \
      acc[1][1] = EHX_MFMA8(A[1][tt], B[1][tt], acc[1][1]);               \
    }                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                    \
  } while (0)

  // MFMA number m (0..15) of a group on fragment set (A,B): tt = m>>2, block = m&3
#define EHX_ONE8(A, B, M)                                                                         \
  acc[(M) & 1][((M) >> 1) & 1] =                                                                  \
      EHX_MFMA8(A[(M) & 1][(M) >> 2], B[((M) >> 1) & 1][(M) >> 2], acc[(M) & 1][((M) >> 1) & 1])

  uint32_t kt = 0, t = 0, buf = 0;
  for (uint32_t step = 0; step < total_steps; ++step) {
    const uint32_t nbuf = buf == kRing8 - 1 ? 0u : buf + 1;
    const char* xs = smem + kXOff8 + buf * kXStage8 + a_row_off;
    const char* qs = smem + kQOff8 + buf * kQStage8 + b_row_off;
    const char* xn = smem + kXOff8 + nbuf * kXStage8 + a_row_off;
    const char* qn = smem + kQOff8 + nbuf * kQStage8 + b_row_off;
    const bool has_next = step + 1 < total_steps;

    // Every MFMA of the stage is emitted exactly once (no alternative code paths around them, so the
    // accumulators stay in place — an earlier version with duplicated MFMA sequences made the
    // compiler copy all 64 accumulator registers between two register sets every stage); only the
    // DMA pieces and the next stage's fragment reads sit behind wave-uniform flags.
    const bool late_dma = late && step >= 1 && issued < total_steps;
    const bool early_dma = !late && has_next && issued < total_steps;

    // ---- group 0 (set 0): the late waves do their DMA duty here, one piece per MFMA ----
    fa1[0] = EHX_XREAD(xs + joffa[1]);
    fa1[1] = EHX_XREAD(xs + kXrb + joffa[1]);
    fb1[0] = frag_read((const char*)(qs + joff[1]));
    fb1[1] = frag_read((const char*)(qs + 4096 + joff[1]));
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      EHX_ONE8(fa0, fb0, m);
      if (m < 7) {  // 7 pieces: wave 0 may be a late wave and owns the row-parameter piece
        if (late_dma) EHX_PIECE(m);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (late_dma) EHX_STAGE_ADVANCE();
    EHX_GROUP8(fa1, fb1, fa0, fb0, xs + joffa[2], qs + joff[2]);
    EHX_GROUP8(fa0, fb0, fa1, fb1, xs + joffa[3], qs + joff[3]);

    // ---- group 3 (set 1): 4 MFMAs, the stage barrier, then 12 MFMAs with the next stage's first
    // fragment reads and (early waves) the DMA pieces of stage step+3 in their shadow ----
#pragma unroll
    for (int m = 0; m < 4; ++m) EHX_ONE8(fa1, fb1, m);
    __builtin_amdgcn_sched_barrier(0);
    if (has_next) {
      // own pieces of stage step+1 landed?  Only stage step+2's may still be in flight.
      if (step + 2 < total_steps) {
        if (w == 0) wait_vmcnt<kP + 1>();
        else wait_vmcnt<kP>();
      } else {
        wait_vmcnt<0>();
      }
      if (!ABL_NO_BARRIER) hot_barrier();  // B_{step+1}: stage step+1 visible; ring slot `buf` is free again
    }
#pragma unroll
    for (int m = 4; m < 16; ++m) {
      EHX_ONE8(fa1, fb1, m);
      if (has_next) {
        if (m == 4) fa0[0] = EHX_XREAD(xn + joffa[0]);
        if (m == 5) fa0[1] = EHX_XREAD(xn + kXrb + joffa[0]);
        if (m == 6) fb0[0] = frag_read((const char*)(qn + joff[0]));
        if (m == 7) fb0[1] = frag_read((const char*)(qn + 4096 + joff[0]));
      }
      if (m >= 8 && m < 15) {
        if (early_dma) EHX_PIECE(m - 8);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (early_dma) EHX_STAGE_ADVANCE();
    buf = nbuf;
    if (++kt == ktiles) {
      kt = 0;
      if (!ABL_NO_EPILOGUE) epilogue(t);
      else {
        for (int rb = 0; rb < 2; ++rb)
          for (int cb = 0; cb < 2; ++cb) asm volatile("" ::"v"(acc[rb][cb]));
      }
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[rb][cb][r] = 0.0f;
      ++t;
    }
  }
#undef EHX_GROUP8
#undef EHX_XREAD
#undef EHX_ONE8
#undef EHX_PIECE
#undef EHX_STAGE_ADVANCE

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

hipError_t launch_flat_scan8(const ScanArgs& a, hipStream_t st) {
  static DynLdsAttr attr;
  const void* fns[2] = {(const void*)flat_scan8_kernel<false>, (const void*)flat_scan8_kernel<true>};
  if (hipError_t e = attr.ensure(fns, 2, kLdsBytes8); e != hipSuccess) return e;
  const uint32_t grid = a.q_tiles * a.n_chunks;
  if (a.x_half) hipLaunchKernelGGL(flat_scan8_kernel<true>, dim3(grid), dim3(kThreads8), kLdsBytes8, st, a);
  else hipLaunchKernelGGL(flat_scan8_kernel<false>, dim3(grid), dim3(kThreads8), kLdsBytes8, st, a);
  return hipGetLastError();
}

}  // namespace ehx
