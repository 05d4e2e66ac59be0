// flat_scan16_kernel — the exhaustive scan as an fp16-MFMA FILTER (v_mfma_f32_32x32x16_f16, 16x the
// fp32 matrix rate) in front of the canonical fp32 re-rank.  Results stay those of the exact fp32
// scan: the filter only decides which rows become candidates, and it does so with a LOWER BOUND of
// every row's score, so a true top-k row can never be filtered out without the re-rank noticing
// (rerank_kernel certifies each query; uncertified queries are re-run through the fp32 scan).
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
// Kernel shape.  At 1024 flop/clk/SIMD a 32x32x16 MFMA occupies the matrix pipe for only 32 clocks, so
// what limits the scan is everything that is NOT an MFMA: the instruction stream is built to spend
// ~3-4 other instructions per MFMA.
//   * workgroup = 8 waves (two per SIMD), tile 256 rows x 256 queries; wave (wr, wc) = (w>>2, w&3) owns
//     128 rows x 64 queries = 4x2 MFMA blocks = 128 accumulator registers: a k-step (k = 16) is 6
//     fragment reads (ds_read_b128) for 8 MFMAs;
//   * a stage is 32 halves of k: stage rows are 64 bytes, 16-byte chunk c of row r sits at physical
//     chunk c ^ ((r>>2)&3) (conflict-free b128 fragment reads); X stage 16 KiB + Q stage 16 KiB, ring of
//     4 stages, filled by global->LDS DMA three stages ahead: every wave copies 2 X pieces + 2 Q pieces
//     of 1 KiB per stage (waves 0/1 also the tile's row parameters, once per tile), accounted with a
//     counted s_waitcnt vmcnt before the single raw s_barrier per stage;
//   * fragments for the next k-step are read while the current one's 8 MFMAs issue.
// Epilogue: phase 1 reduces each 32x32 block to one min per lane and tests it against the list
// threshold — no per-score bookkeeping; only blocks in which some lane hits (rare once the sample
// passes have set the thresholds) get their per-score bit mask built and the hits appended.
#include "ehx_kernels.h"
#include "k_scan_common.h"

namespace ehx {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int kThreads16 = 512;
constexpr uint32_t kRing16 = 4;
constexpr uint32_t kBK16 = 32;                                       // halves per stage row (64 B)
constexpr uint32_t kRowB16 = 64;                                     // bytes per stage row
constexpr uint32_t kXStage16 = kTileRows16 * kRowB16;                // 16 KiB
constexpr uint32_t kQStage16 = kTileQ * kRowB16;                     // 16 KiB
#if (EHX_ABL & 1024)
constexpr uint32_t kQOff16 = 0;
constexpr uint32_t kXOff16 = kRing16 * kQStage16;
#else
constexpr uint32_t kXOff16 = 0;
constexpr uint32_t kQOff16 = kRing16 * kXStage16;                    // 64 KiB
#endif
constexpr uint32_t kThrKeyOff16 = 2 * kRing16 * kQStage16;            // u64 thr_key[512]
constexpr uint32_t kThrFOff16 = kThrKeyOff16 + kLists8 * 8;          // f32 thr_f[512]
constexpr uint32_t kCntOff16 = kThrFOff16 + kLists8 * 4;             // i32 cnt[512]
constexpr uint32_t kRowpOff16 = kCntOff16 + kLists8 * 4;             // float2 rowp_lds[4][256]
constexpr uint32_t kLdsBytes16 = kRowpOff16 + 4 * kTileRows16 * 8;
static_assert(kLdsBytes16 <= 160 * 1024, "LDS budget");
static_assert(kXStage16 == kQStage16, "the stage loop indexes both rings with one offset");

// Ablation hooks for profiling builds only (scripts/ablate_scan.sh); the shipped library defines none.
#ifndef EHX_ABL
#define EHX_ABL 0
#endif
#define ABL16_NO_EPILOGUE (EHX_ABL & 1)
#define ABL16_NO_DMA (EHX_ABL & 2)
#define ABL16_NO_LDSREAD (EHX_ABL & 4)
#define ABL16_NO_BARRIER (EHX_ABL & 8)
#define ABL16_NO_MFMA (EHX_ABL & 16)
#define ABL16_NO_VMWAIT (EHX_ABL & 32)

__device__ __forceinline__ f16x8 frag16(const char* p) {
#if (EHX_ABL & 4)
  f16x8 v = {1, 2, 3, 4, 5, 6, 7, 8};
  asm volatile("" : "+v"(v));
  return v;
#elif (EHX_ABL & 128)
  // issue the read, never use its result: the MFMAs run on constants
  f16x8 r = *(const volatile f16x8*)p;
  asm volatile("" ::"v"(r));
  f16x8 v = {1, 2, 3, 4, 5, 6, 7, 8};
  asm volatile("" : "+v"(v));
  return v;
#else
  return *(const f16x8*)p;
#endif
}

__device__ __forceinline__ f16x8 frag16_const() {
  f16x8 v = {1, 2, 3, 4, 5, 6, 7, 8};
  asm volatile("" : "+v"(v));
  return v;
}
#if (EHX_ABL & 256)
#define frag16a(P) frag16_const()
#else
#define frag16a(P) frag16(P)
#endif
#if (EHX_ABL & 512)
#define frag16b(P) frag16_const()
#else
#define frag16b(P) frag16(P)
#endif
#if (EHX_ABL & 16)
__device__ __forceinline__ f32x16 EHX_MFMA16(f16x8 a, f16x8 b, f32x16 c) {
  asm volatile("" : "+v"(c) : "v"(a), "v"(b));
  return c;
}
#else
#define EHX_MFMA16(A, B, C) __builtin_amdgcn_mfma_f32_32x32x16_f16((A), (B), (C), 0, 0, 0)
#endif

// s_waitcnt lgkmcnt(0) as a builtin (the compiler's own wait-count bookkeeping sees it, unlike inline asm)
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_s_waitcnt(0xC07F);  // vmcnt 63, expcnt 7, lgkmcnt 0
  __builtin_amdgcn_s_barrier();
}

}  // namespace

size_t scan16_lds_bytes() { return kLdsBytes16; }

// DUMP: the sample pass — no candidate lists; every score of the scanned tiles is written to
// a.dump[row - tile0*256][q] and sample_select_kernel (k_flat.hip) turns them into starting thresholds.
template <bool COS, bool DUMP>
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
  const float2* rowp_lds = (const float2*)(smem + kRowpOff16);
  uint64_t* cand = a.cand + (size_t)blockIdx.x * ((size_t)kLists8 * kCandSlots);
  unsigned long long* gthr = a.gthr + (size_t)qt * kTileQ;

  {  // 512 threads, 512 lists; start from the query's global threshold (the previous pass set it)
    const int wq = ((tid >> 6) & 3) * 64 + (tid & 63);  // list tid belongs to wave tid>>6, query wc*64 + ql
    const unsigned long long g = gthr[wq];
    thr_key[tid] = g;
    thr_f[tid] = g == kKeyInf ? __builtin_inff() : ordered_to_f32((uint32_t)(g >> 32));
    cnt[tid] = 0;
  }
  // per-query gamma of this lane's two query columns (wc*64 + cb*32 + i31)
  const float gam0 = a.qgamma[(size_t)qt * kTileQ + wc * 64 + i31];
  const float gam1 = a.qgamma[(size_t)qt * kTileQ + wc * 64 + 32 + i31];

  const uint32_t tile_begin = a.tile0 + chunk * a.tiles_per_chunk;
  uint32_t tile_end = tile_begin + a.tiles_per_chunk;
  if (tile_end > a.tile0 + a.n_tiles) tile_end = a.tile0 + a.n_tiles;
  const uint32_t my_tiles = tile_end > tile_begin ? tile_end - tile_begin : 0u;
  const uint32_t ktiles = a.ld / kBK16;

  // ---- DMA duty of this wave: 1-KiB pieces w and w+8 of the X stage block and of the Q stage block.
  // The blocks are stored in HBM in the LDS image (scan16_index), so a piece is a linear copy: lane L
  // moves 16 bytes at piece*1024 + L*16.  Sources are uniform 64-bit pointers (SGPR pair) advanced by one
  // 16-KiB block per stage: X runs linearly through its [tile][stage] blocks — past the end of the chunk
  // for the three stages issued ahead at the very end (valid memory: the next chunk's tiles or the
  // buffer's tail padding, never read from LDS) — and Q restarts at stage 3 of its tile every tile, its
  // tile being stored with stages 0..2 repeated after the last one.
  const uint32_t voff = (uint32_t)lane * 16u;
  const uint32_t voff8 = voff + 8u * 1024u;
  const size_t tile_bytes = (size_t)ktiles * kXStage16;
  const char* xsrc = (const char*)a.X + (size_t)tile_begin * tile_bytes + (size_t)w * 1024;
  const char* qbase = (const char*)a.Q + (size_t)qt * ((size_t)(ktiles + 3) * kQStage16) + (size_t)w * 1024;
  const char* qsrc = qbase;
  const char* rsrc = (const char*)(a.rowp + (size_t)tile_begin * kTileRows16) + (size_t)(w & 1) * 1024;
  const uint32_t xdst = kXOff16 + (uint32_t)w * 1024u;  // + slot*16384 (+8192 for the second piece)
  const uint32_t qdst = kQOff16 + (uint32_t)w * 1024u;
  const uint32_t rdst = kRowpOff16 + (uint32_t)(w & 1) * 1024u;  // + (tile&3)*2048

  // one 1-KiB piece: M0 = LDS byte address of the piece, source = uniform base + per-lane offset
#define EHX_DMA(DST_BASE, DST_IMM, VOFF, SRC)                                                              \
  do {                                                                                                     \
    if (!ABL16_NO_DMA)                                                                                     \
      asm volatile("s_add_u32 m0, %0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3"                     \
                   :                                                                                       \
                   : "s"(DST_BASE), "n"(DST_IMM), "v"(VOFF), "s"(SRC)                                      \
                   : "memory", "scc");                                                                     \
  } while (0)
  // this wave's four pieces of the next stage, into ring slot SLOT
#define EHX_DMA_X0(SLOT) EHX_DMA(xdst, (SLOT) * 16384, voff, xsrc)
#define EHX_DMA_Q0(SLOT) EHX_DMA(qdst, (SLOT) * 16384, voff, qsrc)
#define EHX_DMA_X1(SLOT) EHX_DMA(xdst, (SLOT) * 16384 + 8192, voff8, xsrc)
#define EHX_DMA_Q1(SLOT)                              \
  do {                                                \
    EHX_DMA(qdst, (SLOT) * 16384 + 8192, voff8, qsrc); \
    if (!(EHX_ABL & 64)) {                            \
      xsrc += kXStage16;                              \
      qsrc += kQStage16;                              \
    }                                                 \
  } while (0)

  // ---- fragment read offsets: k-step j (0/1) of a stage row is its logical chunks 2j (lanes 0-31) and
  // 2j+1 (lanes 32-63) — any fixed k permutation is fine as long as rows and queries use the same one
  const uint32_t sw = ((uint32_t)i31 >> 2) & 3u;
  const uint32_t a_off0 = kXOff16 + (uint32_t)(wr * 128 + i31) * kRowB16 + (((uint32_t)h) ^ sw) * 16u;       // + rb*2048
  const uint32_t a_off1 = kXOff16 + (uint32_t)(wr * 128 + i31) * kRowB16 + ((2u + (uint32_t)h) ^ sw) * 16u;
  const uint32_t b_off0 = kQOff16 + (uint32_t)(wc * 64 + i31) * kRowB16 + (((uint32_t)h) ^ sw) * 16u;        // + cb*2048
  const uint32_t b_off1 = kQOff16 + (uint32_t)(wc * 64 + i31) * kRowB16 + ((2u + (uint32_t)h) ^ sw) * 16u;

  f32x16 acc[4][2];
  const float acc0 = a.eps;  // accumulators start at +eps: the scores are lower bounds
#pragma unroll
  for (int rb = 0; rb < 4; ++rb)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[rb][cb][r] = acc0;

  // =============================== tile epilogue ===============================
  auto epilogue = [&](uint32_t t) {
    const uint32_t tile_row0 = (tile_begin + t) * kTileRows16;
    const float2* rp = rowp_lds + (t & 3u) * kTileRows16;
    if (DUMP) {
      const size_t qcol = (size_t)qt * kTileQ + wc * 64 + i31;
      const size_t q_rows = (size_t)a.q_tiles * kTileQ;
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const uint32_t r = (uint32_t)(wr * 128 + rb * 32 + (reg & 3) + 8 * (reg >> 2)) + 4u * h;
          const float2 ab = rp[r];
          float* o = a.dump + (size_t)(tile_row0 - a.tile0 * kTileRows16 + r) * q_rows + qcol;
          o[0] = __builtin_fmaf(ab.x, acc[rb][0][reg], ab.y * gam0);
          o[32] = __builtin_fmaf(ab.x, acc[rb][1][reg], ab.y * gam1);
        }
      }
      return;
    }
    const int lbase = w * 64 + i31;  // + cb*32
    const float thrf0 = thr_f[lbase], thrf1 = thr_f[lbase + 32];
    // ---- phase 1: one min score per 32x32 block and lane ----
    float bm[4][2];
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
      if (COS) {
        // valid rows have (a, b) = (-1, 1) and gamma = 1: min S = 1 - max acc, no row parameters needed
        // (padding / invalid rows may raise a false alarm; phase 2 evaluates them properly)
        float m0 = acc[rb][0][0], m1 = acc[rb][1][0];
#pragma unroll
        for (int reg = 1; reg < 16; ++reg) {
          m0 = fmaxf(m0, acc[rb][0][reg]);
          m1 = fmaxf(m1, acc[rb][1][reg]);
        }
        bm[rb][0] = __builtin_fmaf(-1.0f, m0, 1.0f);
        bm[rb][1] = __builtin_fmaf(-1.0f, m1, 1.0f);
      } else {
        float m0 = __builtin_inff(), m1 = __builtin_inff();
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const uint32_t r = (uint32_t)(wr * 128 + rb * 32 + (reg & 3) + 8 * (reg >> 2)) + 4u * h;
          const float2 ab = rp[r];
          m0 = fminf(m0, __builtin_fmaf(ab.x, acc[rb][0][reg], ab.y * gam0));
          m1 = fminf(m1, __builtin_fmaf(ab.x, acc[rb][1][reg], ab.y * gam1));
        }
        bm[rb][0] = m0;
        bm[rb][1] = m1;
      }
    }
    const float mm0 = fminf(fminf(bm[0][0], bm[1][0]), fminf(bm[2][0], bm[3][0]));
    const float mm1 = fminf(fminf(bm[0][1], bm[1][1]), fminf(bm[2][1], bm[3][1]));
    if (!__any((mm0 <= thrf0) | (mm1 <= thrf1))) return;  // common case once the thresholds are warm
    // ---- slow path, block by block: per-score bit mask (bit = reg), then the appends ----
    const int trigger = (int)a.kprime + ((int)kCandSlots - (int)a.kprime) / 2;
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        const int list = lbase + cb * 32;
        if (!__any(bm[rb][cb] <= thr_f[list])) continue;
        const float gam = cb ? gam1 : gam0;
        const uint32_t rbase = (uint32_t)(wr * 128 + rb * 32) + 4u * h;
        uint32_t pend = 0u;
        {
          const float thrf = thr_f[list];
#pragma unroll
          for (int reg = 0; reg < 16; ++reg) {
            const float2 ab = rp[rbase + (reg & 3) + 8 * (reg >> 2)];
            const float s = __builtin_fmaf(ab.x, acc[rb][cb][reg], ab.y * gam);
            pend |= (s <= thrf) ? (1u << reg) : 0u;
          }
        }
        for (int round = 0; round < 1024; ++round) {
          bool hit_trigger = false;
          uint32_t retry = 0u;
          while (__any(pend != 0u)) {
            if (pend != 0u) {
              const int b = __builtin_ctz(pend);
              pend &= pend - 1u;
              float dot = 0.0f;
#pragma unroll
              for (int i = 0; i < 16; ++i) dot = (b == i) ? acc[rb][cb][i] : dot;
              const uint32_t r = rbase + (b & 3) + 8 * (b >> 2);
              const float2 ab = rp[r];
              const float sc = __builtin_fmaf(ab.x, dot, ab.y * gam);
              const int pos = scan8_push(sc, tile_row0 + r, list, a.n, cand, cnt, thr_key);
              if (pos >= (int)kCandSlots) retry |= 1u << b;  // list full: compact, then try again
              hit_trigger |= pos + 1 >= trigger;
            }
          }
          pend = retry;
          if (!__any(hit_trigger)) break;
          scan8_compact(w, wc, lane, (int)a.kprime, false, cand, cnt, thr_key, thr_f, gthr);
          if (!__any(pend != 0u)) break;
          if (round == 1023 && lane == 0) atomicAdd(a.err, 1u);  // never reached: a compacted list has free slots
        }
      }
    }
  };

  __syncthreads();  // state init visible
  if (my_tiles > 0) {  // (a chunk past the end of the pass has nothing to scan and must not touch memory)
  // ---- prologue: row parameters of tile 0, stages 0..2 into ring slots 0..2 ----
  if (w < 2) EHX_DMA(rdst, 0, voff, rsrc);
  EHX_DMA_X0(0); EHX_DMA_Q0(0); EHX_DMA_X1(0); EHX_DMA_Q1(0);
  EHX_DMA_X0(1); EHX_DMA_Q0(1); EHX_DMA_X1(1); EHX_DMA_Q1(1);
  EHX_DMA_X0(2); EHX_DMA_Q0(2); EHX_DMA_X1(2); EHX_DMA_Q1(2);
  wait_vmcnt<8>();  // stage 0 (and the row parameters, older) landed <=> at most stages 1, 2 in flight
  lds_barrier();  // B_0

  // Fragments are double-buffered: set 0 feeds k-step 0, set 1 feeds k-step 1; the six fragment reads of
  // the next k-step are issued ahead of the current k-step's 8 MFMAs and land in their shadow.
  f16x8 fa0[4], fb0[2], fa1[4], fb1[2];
#pragma unroll
  for (int rb = 0; rb < 4; ++rb) fa0[rb] = frag16a(smem + a_off0 + rb * 2048);
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) fb0[cb] = frag16b(smem + b_off0 + cb * 2048);

  // One stage, ring slot S (compile-time): no branches, no address arithmetic — fragment reads are
  // base register + immediate, the DMA of stage +3 goes to slot S+3, one counted wait + one barrier.
#define EHX_STAGE16(S)                                                                                   \
  do {                                                                                                   \
    constexpr uint32_t so = (uint32_t)(S) * kXStage16, sn = (uint32_t)(((S) + 1) & 3) * kXStage16;       \
    constexpr int sd = ((S) + 3) & 3;                                                                    \
    /* Every gap between two MFMAs carries exactly one other instruction of this wave (a fragment read or \
       a DMA piece): bunched together they would leave the matrix pipe idle while they issue. */         \
    /* k-step 0: MFMAs on set 0; reads of set 1 (k-step 1 of this stage); first two DMA pieces */        \
    EHX_MF(fa0, fb0, 0, 0); fb1[0] = frag16b(smem + b_off1 + so);            EHX_SB();                    \
    EHX_MF(fa0, fb0, 0, 1); fb1[1] = frag16b(smem + b_off1 + so + 2048);     EHX_SB();                    \
    EHX_MF(fa0, fb0, 1, 0); fa1[0] = frag16a(smem + a_off1 + so);            EHX_SB();                    \
    EHX_MF(fa0, fb0, 1, 1); fa1[1] = frag16a(smem + a_off1 + so + 2048);     EHX_SB();                    \
    EHX_MF(fa0, fb0, 2, 0); fa1[2] = frag16a(smem + a_off1 + so + 4096);     EHX_SB();                    \
    EHX_MF(fa0, fb0, 2, 1); fa1[3] = frag16a(smem + a_off1 + so + 6144);     EHX_SB();                    \
    EHX_MF(fa0, fb0, 3, 0); EHX_DMA_X0(sd);                                 EHX_SB();                    \
    EHX_MF(fa0, fb0, 3, 1); EHX_DMA_Q0(sd);                                 EHX_SB();                    \
    /* stage barrier: the next stage landed (own pieces counted: the younger stage and the two pieces    \
       just issued may still be in flight) and is visible; every wave is done reading this stage */      \
    if (!ABL16_NO_VMWAIT) wait_vmcnt<6>();                                                               \
    if (!ABL16_NO_BARRIER) lds_barrier();                                                                \
    EHX_SB();                                                                                            \
    /* k-step 1: MFMAs on set 1; reads of set 0 of the next stage; last two DMA pieces */                \
    EHX_MF(fa1, fb1, 0, 0); fb0[0] = frag16b(smem + b_off0 + sn);            EHX_SB();                    \
    EHX_MF(fa1, fb1, 0, 1); fb0[1] = frag16b(smem + b_off0 + sn + 2048);     EHX_SB();                    \
    EHX_MF(fa1, fb1, 1, 0); fa0[0] = frag16a(smem + a_off0 + sn);            EHX_SB();                    \
    EHX_MF(fa1, fb1, 1, 1); fa0[1] = frag16a(smem + a_off0 + sn + 2048);     EHX_SB();                    \
    EHX_MF(fa1, fb1, 2, 0); fa0[2] = frag16a(smem + a_off0 + sn + 4096);     EHX_SB();                    \
    EHX_MF(fa1, fb1, 2, 1); fa0[3] = frag16a(smem + a_off0 + sn + 6144);     EHX_SB();                    \
    EHX_MF(fa1, fb1, 3, 0); EHX_DMA_X1(sd);                                 EHX_SB();                    \
    EHX_MF(fa1, fb1, 3, 1); EHX_DMA_Q1(sd);                                 EHX_SB();                    \
  } while (0)
#define EHX_MF(A, B, RB, CB) acc[RB][CB] = EHX_MFMA16(A[RB], B[CB], acc[RB][CB])
#define EHX_SB() __builtin_amdgcn_sched_barrier(0)

  // One flat loop over groups of four stages (= one revolution of the ring; ld % 128 == 0 makes a tile a
  // whole number of them); the tile boundary work hangs off a counter inside it.
  const uint32_t kquads = ktiles >> 2;
  const uint32_t total_quads = my_tiles * kquads;
  // row parameters of tile 1 (consumed by its epilogue, a whole tile from now)
  rsrc += kTileRows16 * 8;
  if (w < 2) EHX_DMA(rdst, 2048, voff, rsrc);
  uint32_t kq = 0, t = 0;
  for (uint32_t q = 0; q < total_quads; ++q) {
    EHX_STAGE16(0);
    EHX_STAGE16(1);
    EHX_STAGE16(2);
    EHX_STAGE16(3);
    if (++kq == kquads) {
      kq = 0;
      if (!ABL16_NO_EPILOGUE) epilogue(t);
      else {
        for (int rb = 0; rb < 4; ++rb)
          for (int cb = 0; cb < 2; ++cb) asm volatile("" ::"v"(acc[rb][cb]));
      }
#pragma unroll
      for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[rb][cb][r] = acc0;
      ++t;
      // next tile: its stage 3 is the next one to issue (stages 0..2 came from the repeated blocks); row
      // parameters of the tile after it (past the last tile: the array's two tiles of tail padding)
      qsrc = qbase + 3 * kQStage16;
      rsrc += kTileRows16 * 8;
      if (w < 2) {
        const uint32_t rd = rdst + ((t + 1) & 3u) * 2048u;
        EHX_DMA(rd, 0, voff, rsrc);
      }
    }
  }
  }  // my_tiles > 0
#undef EHX_STAGE16
#undef EHX_MF
#undef EHX_SB
#undef EHX_DMA_X0
#undef EHX_DMA_Q0
#undef EHX_DMA_X1
#undef EHX_DMA_Q1
#undef EHX_DMA

  // ---- final: sort this wave's 64 lists and publish them: part[q][chunk*2 + wr][k'] ----
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  if (DUMP) return;
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
  static DynLdsAttr attr;
  const void* fns[3] = {(const void*)flat_scan16_kernel<false, false>, (const void*)flat_scan16_kernel<true, false>,
                        (const void*)flat_scan16_kernel<false, true>};
  if (hipError_t e = attr.ensure(fns, 3, kLdsBytes16); e != hipSuccess) return e;
  const uint32_t grid = a.q_tiles * a.n_chunks;
  if (a.dump) hipLaunchKernelGGL((flat_scan16_kernel<false, true>), dim3(grid), dim3(kThreads16), kLdsBytes16, st, a);
  else if (a.cos) hipLaunchKernelGGL((flat_scan16_kernel<true, false>), dim3(grid), dim3(kThreads16), kLdsBytes16, st, a);
  else hipLaunchKernelGGL((flat_scan16_kernel<false, false>), dim3(grid), dim3(kThreads16), kLdsBytes16, st, a);
  return hipGetLastError();
}

}  // namespace ehx
