// flat_scan_i8_kernel — the exhaustive scan as an INT8 matrix-core FILTER (v_mfma_i32_16x16x64_i8 since round 4, 32x32x32
// before: twice the fp16 matrix rate on half the bytes, EXACT int32 accumulation) in front of the canonical fp32 re-rank.
// Like the fp16 filter (k_flat16.hip) it only decides which rows become candidates, with a certified LOWER BOUND of every
// row's score, so the answer stays the exact fp32 answer (rerank256_kernel certifies each query; what it cannot
// certify falls to the fp16 filter, the fp32 scan and finally the exhaustive canonical pass).
//
// The scan copy (k_misc.hip: make_scan8): every row normalised to unit length (x^ = x / n_r) and quantised with
// its own scale,  x^ = s_r * xi + dx,  xi int8,  s_r >= max|x^| / 127,  e_r >= |dx|;  queries likewise
// (s_q, qi, e_q).  The integer dot product I = <qi, xi> is exact, so with dot^ = <q^, x^>
//     dot^  <=  U = s_q s_r I + e_q (1.0001 + e_r) + 1.0001 e_r          (Cauchy-Schwarz on the two residuals)
// and every metric of the engine is affine in dot^ with a non-positive slope (table in k_flat16.hip):
//     S(r,q) = b_r*gamma_q + a_r*dot^  >=  S_lower = B_r*gamma_q + D_r + C_r*e_q + A_r*(s_q*I)
// with the row parameters (A, B, C, D) = (a_r s_r, b_r(1-1e-6), a_r(1.0001 + e_r), a_r(1.0001 e_r + slack)) computed
// when the row is written (slack: the fp32 evaluation of this expression, the filter's own norms — make_scan8).
//
// Kernel shape: workgroup = 8 waves (two per SIMD), tile 256 rows x 256 queries, wave tile 128 x 64 = 8 x 4 MFMA blocks
// of 16 x 16; a stage row is 64 bytes = ONE k-step of 64, X stage 16 KiB + Q stage 16 KiB, ring of 4 stages filled
// three stages ahead by global->LDS DMA from the stage-blocked scan copy; one counted s_waitcnt + one raw s_barrier per
// stage (details at the kernel).
//
// Short rows (round 4; a tile of at most four stages: ld <= 192 in the run-time-slot loop): the query tile's stage
// blocks are the same for every row tile, so they are copied into the Q ring's four slots ONCE and stay there (QRES);
// a stage then copies its two X pieces per wave only — half the DMA instructions (each costs 60-180 cycles of issue
// time among MFMAs) and half the L2 -> LDS bytes.  EHX_I8_QRES=0 in the environment: the ring for both, as before.
//
// Epilogue.  The threshold of a (query, pass) is FIXED — the k'-th best lower bound of the rows scanned by the earlier
// passes (select256_kernel) — and a pass collects every (row, query) whose lower bound is not above it:
//   phase 1  per query of the lane (four per lane), ONE integer maximum over the lane's 32 accumulators against ONE
//            integer level: the rows of a full tile are stored ordered by quantisation step and share one |A| per
//            32-row lane group (k_misc.hip), so I |A_r| >= K_q is I >= floor(K_q / |A|), exactly; K_q comes from the
//            tile's parameter extremes (tile_params8), once per wave, handed round by a lane permute;
//   phase 2  (a lane reached its level) the row blocks whose own maximum reaches it, their accumulators as a 32-bit
//            mask per lane; each trip of a short loop takes one set bit per lane, evaluates S_lower with that row's
//            parameters (the one LDS round trip of the path) and stages (score, position, query) in the wave's own
//            buffer in LDS (128 entries; the fill count is a wave-uniform register, slots by ballot + lane prefix);
//   a staging buffer that runs full is flushed to the queries' POOLS in HBM (one global atomicAdd per entry, the
//   position mapped to the row id through perm8); what is left is flushed when the workgroup is done.  The hot path
//   touches LDS only: a global store or atomic per hit would put foreign entries into the vmcnt queue the stage loop
//   counts on (loads and stores may retire out of order with respect to each other), so the flush — rare — drains the
//   queue instead.  A pool that overflows (adversarial row order: every row beats a stale threshold) only flags its
//   query — the next engine of the chain answers it.
// A cascade of passes x4 in rows keeps the hits at ~2-3 k' per query and pass.
//
// Lock-step (optional, ScanArgsI8::sync / sync_tol, EHX_I8_SYNC=N): the q_tiles workgroups that stream the same row chunk
// sit on one XCD and share its L2, but nothing keeps them together, and once they drift apart by more than the L2 holds
// every one of them fetches the rows from HBM again (1.13 x the algorithmic traffic at 10 M x 768).  By TILE: each
// workgroup publishes how many tiles it has finished in a word of its own and does not run more than N tiles ahead of
// its slowest sibling (details at after_tile).  1.04 x the traffic for +1.3 % time (profiles/r05_m_*); the default is
// decided by measurement with something else on the bus.  (Rounds 2-4 synchronised by ring REVOLUTION with
// one shared counter: superseded, removed in round 6 — profiles/r05_l_sync.jsonl is its record.)
#include "ehx_env.h"
#include "ehx_kernels.h"
#include "k_scan_common.h"

namespace ehx {

typedef int i32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr uint32_t kRingI8 = 4;
constexpr uint32_t kRowBI8 = 64;                                   // bytes (= k-values) per stage row
constexpr uint32_t kStageI8 = kTileRows16 * kRowBI8;               // 16 KiB: one (tile, stage) block in HBM, X and Q alike
// LDS layout and geometry of one workgroup.  HALF = false: 8 waves, 256 rows x 256 queries (everything above).
// HALF = true (round 5, short rows): 4 waves — ONE wave per SIMD and workgroup — take one 128-row HALF of every tile of
// their chunk against the resident query tile; two such workgroups share a CU (<= 80 KiB of LDS each), so one's tile
// epilogue (vector ALU only) runs beside the other's matrix work instead of all eight waves of a CU leaving the matrix
// pipe together.
template <bool HALF>
struct I8L {
  static constexpr int kThreads = HALF ? 256 : 512;
  static constexpr uint32_t kWaves = HALF ? 4 : 8;
  static constexpr uint32_t kXStage = HALF ? kStageI8 / 2 : kStageI8;  // X bytes of one stage in LDS (128 / 256 rows x 64 B)
  static constexpr uint32_t kXShift = HALF ? 13 : 14;
  static constexpr uint32_t kQSlots = HALF ? 2 : kRingI8;             // 16-KiB query stage blocks held in LDS
  static constexpr uint32_t kStgCap = HALF ? 64 : 128;                // staging entries per wave
  static constexpr uint32_t kRowpSlot = (HALF ? 128 : 256) * 16;      // row parameters of one (half) tile
  static constexpr uint32_t kRowpWaves = HALF ? 2 : 4;                // waves that copy a 1-KiB piece of them
  static constexpr uint32_t kXOff = 0;
  static constexpr uint32_t kQOff = kRingI8 * kXStage;                // 64 KiB (HALF: 32)
  static constexpr uint32_t kRowpOff = kQOff + kQSlots * kStageI8;    // float4 rowp_lds[3][256 (128)]
  static constexpr uint32_t kQpOff = kRowpOff + 3 * kRowpSlot;        // float4 qp_lds[256] = (s_q, e_q, gamma_q, thr_q)
  static constexpr uint32_t kQinvOff = kQpOff + kTileQ * 16;          // float qinv_lds[256] = (1 - 1e-5) / s_q
  static constexpr uint32_t kSyncOff = kQinvOff + kTileQ * 4;         // u32 snapshot[64] of the lock-step counter
  static constexpr uint32_t kCtxOff = kSyncOff + 256;                 // I8Ctx: what the (rare) flush path needs
  static constexpr uint32_t kStgKeyOff = kCtxOff + 64;                // u64 stg_key[waves][cap]
  static constexpr uint32_t kStgQlOff = kStgKeyOff + kWaves * kStgCap * 8;  // u32 stg_ql[waves][cap]
  static constexpr uint32_t kStgCntOff = kStgQlOff + kWaves * kStgCap * 4;  // u32 stg_cnt[8]
  static constexpr uint32_t kLdsBytes = kStgCntOff + 64;
};
static_assert(I8L<false>::kLdsBytes <= 160 * 1024, "LDS budget");
static_assert(I8L<true>::kLdsBytes <= 80 * 1024, "LDS budget: two half-tile workgroups per CU");
static_assert(I8L<false>::kQOff == 64 * 1024 && I8L<false>::kRowpOff == 128 * 1024, "layout of the full-tile kernel");
static_assert(kTileRows16 == 256 && kTileQ == 256, "kernel geometry is hard-wired");

#define EHX_MFMA_I8(A, B, C) __builtin_amdgcn_mfma_i32_16x16x64_i8((A), (B), (C), 0, 0, 0)

// Ablation builds (scripts/ablate_i8.sh; results are WRONG by construction, only the scan's duration is looked at):
//   1  no tile epilogue      2  epilogue phase 1 only (alarms never taken)      4  no DMA after the prologue (the ring
//   keeps its first three stages; no counted wait)      8  no fragment reads (MFMAs on the prologue's fragments)
//   16 no stage barrier                               32 no epilogue arithmetic (accumulators kept alive: 1 lets the
//                                                          compiler delete the MFMAs)
#ifndef EHX_I8_ABL
#define EHX_I8_ABL 0
#endif
// EHX_I8_COUNT: diagnosis build — a.cand[0..4] count (phase-1 tests, alarms, alarmed row blocks, phase-2 trips, staged
// keys) per launch sequence (never in the shipped library)
#ifndef EHX_I8_COUNT
#define EHX_I8_COUNT 0
#endif
// (Round 4 built a FUSED epilogue — tile t-1's alarm tests inside the first stage of tile t, between its MFMAs: bit-identical
// and 1-5 % slower on every shape (profiles/r04_v_ab_flat.jsonl: the epilogue's instructions do not hide behind the MFMAs of
// their own SIMD, and where the clock is power-limited overlap cannot shorten an energy bill).  Removed in round 6.)
// Static issue priority for the second-resident wave of every SIMD (-DEHX_I8_PRIO=0: A/B builds without it).  Same-box
// ABAB, 40 batches each, scan time per batch (profiles/r06_i_prio_ab.jsonl): 6.25 M x 128 1.0103 / 1.0123 -> 1.0002 /
// 0.9911 ms (-1.5 %), 10 M x 768 5.9394 / 5.9117 -> 5.9103 / 5.9066 (-0.3 %), 1 M x 768 0.7624 / 0.7652 -> 0.7616 / 0.7644;
// identical id checksums.
#ifndef EHX_I8_PRIO
#define EHX_I8_PRIO 1
#endif
#if EHX_I8_COUNT
#define EHX_CNT(I) do { if (lane == 0) atomicAdd((unsigned long long*)a.cand + (I), 1ull); } while (0)
#else
#define EHX_CNT(I) do { } while (0)
#endif

__device__ __forceinline__ void lds_barrier_i8() {
  __builtin_amdgcn_s_waitcnt(0xC07F);  // vmcnt 63, expcnt 7, lgkmcnt 0
  __builtin_amdgcn_s_barrier();
}

// S_lower of one (row, query) from the row parameters P = (A, B, C, D), the query parameters
// qq = (s_q, e_q, gamma_q, -) and the exact integer dot product v
__device__ __forceinline__ float i8_score(const float4 P, const float4 qq, int v) {
  const float t = qq.x * (float)v;
  const float K = __builtin_fmaf(P.y, qq.z, __builtin_fmaf(P.z, qq.y, P.w));
  return __builtin_fmaf(P.x, t, K);
}

// Alarm threshold of a tile for one query, on the PRODUCT I * |A_r| (exact integer dot product times the row's own
// step): an accumulator can only belong to a row with S_lower <= thr if  I * |A_r| >= K.  From
//   S_lower(r) = B gamma + D + C e_q - |A_r| s_q I  >=  Bmin*gamma - Dmax - Cmax*e_q - |A_r| s_q I
// with tp = (max|A|, max|C|, max|D|, min B) over the tile's rows (tile_params8):  K = (Bmin gamma - Dmax - Cmax e_q -
// thr) / s_q.  The first version compared I against ONE integer per (tile, query), which has to assume the tile's
// largest step: max|x/|x|| varies +-10 % between rows and the largest of 256 is 1.37 x the mean, so on 768-dim Gaussian
// rows the level sat at 2.6 sigma instead of 3.6 and 99 % of the 32 x 32 blocks took the slow path for 0.14 candidates
// per block (tests/test_i8_model.py reproduces the rates).  Per row the test costs a convert and a multiply per
// accumulator and 18-50 % of the blocks go on (scripts/studies/int8_alarm_rates.py).  K errs towards alarms
// (relative slack); -inf = always, +inf = never.
__device__ __forceinline__ float i8_alarm_k(const float4 tp, const float4 qq, const float qinv) {
  const float bg = tp.w * qq.z, ce = tp.y * qq.y;
  float num = bg - tp.z - ce - qq.w;
  num -= 1e-5f * (fabsf(bg) + tp.z + ce + fabsf(qq.w));
  if (!(num > 0.0f)) return -__builtin_inff();              // thr = +inf, NaN, or the bound is already below thr
  return num * qinv;   // qinv = (1 - 1e-5) / s_q, or +inf when the query's step is 0 (S_lower does not depend on I)
}

// What the flush path needs lives in LDS (written once per workgroup): the out-of-line flush takes no arguments.
struct I8Ctx {
  uint64_t* pool;        // [q_rows][pool_cap]
  uint32_t* pool_cnt;    // [q_rows]
  uint32_t* ovf;         // [q_rows]
  const uint8_t* perm;   // [cap] position -> row index inside its tile (tiles ordered by step: k_misc.hip)
  const float4* rowp;    // [cap + 512] row parameters by POSITION (the flush evaluates the staged hits' lower bounds)
  uint32_t pool_cap;
  uint32_t n;            // valid rows
  uint32_t q_tile0;      // global index of this workgroup's query 0 (q_tile * 256)
  uint32_t pad;
};
static_assert(sizeof(I8Ctx) <= 64, "I8Ctx slot");

// Empty this wave's staging buffer into the pools of its queries (wave-uniform call).  A staged entry is a HIT of the
// integer level test — (exact integer dot product, position, query) — not yet a key: the flush evaluates its exact lower
// bound, one entry per lane, drops what lies above the query's threshold, and gives every survivor one slot of its query's
// pool with a global atomicAdd.  (Round 6.  Until then the epilogue evaluated every hit on the spot: an LDS round trip for
// the row's parameters and the score's arithmetic with one or two lanes of 64 at work, while the three (seven) sibling waves
// waited at the next stage barrier — a quarter of the scan time at 6.25 M x 128, profiles/r06_s_*.)  The row parameters come
// from HBM here (the tile's LDS copy is long gone); the vmcnt queue is drained before returning, so the caller's counted
// waits see DMA pieces only.
template <bool HALF>
__device__ __attribute__((noinline)) void i8_flush_staging() {
  using L = I8L<HALF>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const I8Ctx* ctx = (const I8Ctx*)(smem + L::kCtxOff);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const uint64_t* keys = (const uint64_t*)(smem + L::kStgKeyOff) + (size_t)w * L::kStgCap;
  const uint32_t* qls = (const uint32_t*)(smem + L::kStgQlOff) + (size_t)w * L::kStgCap;
  const float4* qp_lds = (const float4*)(smem + L::kQpOff);
  uint32_t* cnt = (uint32_t*)(smem + L::kStgCntOff) + w;
  uint32_t n = *cnt;
  if (n > L::kStgCap) n = L::kStgCap;
  const uint32_t cap = ctx->pool_cap;
  for (uint32_t i = (uint32_t)lane; i < n; i += 64) {
    const uint64_t e = keys[i];
    // the staged id is the row's POSITION in the scan copy; its tile may be stored ordered by step
    const uint32_t posn = (uint32_t)e;
    const int v = (int)(uint32_t)(e >> 32);
    const uint32_t qloc = (uint32_t)(w & 3) * 64u + qls[i];
    const float4 qq = qp_lds[qloc];
    const float S = i8_score(ctx->rowp[posn], qq, v);
    if (!(S <= qq.w) || posn >= ctx->n) continue;
    const uint64_t key = ((uint64_t)f32_to_ordered(S) << 32) | (uint64_t)((posn & 0xFFFFFF00u) | (uint32_t)ctx->perm[posn]);
    const uint32_t q = ctx->q_tile0 + qloc;
    const uint32_t pos = atomicAdd(&ctx->pool_cnt[q], 1u);
    if (pos < cap) ctx->pool[(size_t)q * cap + pos] = key;
    else ctx->ovf[q] = 1u;  // the pool is full: the query is answered by the next engine
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  if (lane == 0) *cnt = 0u;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// phase 2 of the epilogue for one accumulator value per lane: lanes with `hi` (the accumulator reached the integer level)
// stage (value, position, query) for the flush, which judges it.  No read, no wait: the staging buffer belongs to this wave
// alone, so its fill count lives in a wave-uniform register (stg_n) and the slots are handed out by a ballot and a lane
// prefix count.
template <bool HALF>
__device__ __forceinline__ void i8_hit(int v, bool hi, uint32_t r_local, uint32_t tile_row0, int ql, int w, uint32_t& stg_n) {
  using L = I8L<HALF>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const uint64_t m = __ballot(hi);
  if (m == 0ull) return;
  const uint32_t k = (uint32_t)__builtin_popcountll(m);
  uint32_t* cnt = (uint32_t*)(smem + L::kStgCntOff) + w;
  if (stg_n + k > L::kStgCap) {  // (64 lanes, at least 64 entries: an emptied buffer takes them all)
    if ((threadIdx.x & 63) == 0) *cnt = stg_n;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    i8_flush_staging<HALF>();
    stg_n = 0u;
  }
  if (hi) {
    const uint32_t pos = stg_n + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    uint64_t* keys = (uint64_t*)(smem + L::kStgKeyOff) + (size_t)w * L::kStgCap;
    uint32_t* qls = (uint32_t*)(smem + L::kStgQlOff) + (size_t)w * L::kStgCap;
    keys[pos] = ((uint64_t)(uint32_t)v << 32) | (uint64_t)(tile_row0 + r_local);
    qls[pos] = (uint32_t)ql;
  }
  stg_n += k;
}

}  // namespace

size_t scan_i8_lds_bytes() { return I8L<false>::kLdsBytes; }

// DUMP: the sample pass — every lower bound of the scanned tiles is written to a.dump[scan8_dump_index(q, row - tile0*256)] and
// sample_select256_kernel turns them into the first thresholds.
//
// Round 4: the matrix instruction is v_mfma_i32_16x16x64_i8.  Same rate on paper as 32x32x32, but measured on this part
// (scripts/ubench/i8_mfma_shapes.hip, profiles/r04_a_i8_mfma_shapes_ubench.txt; register-resident operands, two waves
// per SIMD): 4200-4280 TOP/s at 2.10-2.17 GHz against 3400-3680 at 1.69-1.88 GHz — the 32x32x32 stream is what makes
// the part clock down.  Wave tile unchanged (128 rows x 64 queries): 8 x 4 blocks of 16 x 16, four accumulator
// registers each; a stage row's 64 bytes are ONE k-step: lane l holds bytes [16 (l >> 4), +16) of row / query l & 15 of
// its block — one ds_read_b128 per block and stage, twelve per stage as before.  Every block gets exactly one MFMA
// per stage.  An accumulator block holds, per lane, rows 4 (l >> 4) + 0..3 of its 16 rows for query l & 15.
template <bool DUMP, bool REV, bool QRES = false, bool HALF = false, bool PAIR = false>
__global__ __launch_bounds__(I8L<HALF>::kThreads, 2) void flat_scan_i8_kernel(const ScanArgsI8 a) {
  static_assert(!QRES || (!REV && !DUMP), "QRES: the run-time-slot loop of the plain scan only");
  static_assert(!PAIR || (!REV && !DUMP && !QRES && !HALF), "PAIR: the run-time-slot loop of the plain scan, stages in pairs");
  static_assert(!HALF || QRES, "HALF: short rows, the query tile resident in LDS");
  using L = I8L<HALF>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j15 = lane & 15, qd = lane >> 4;

  uint32_t qt, chunk, half = 0u;   // half (HALF): which 128 rows of every tile this workgroup takes
  {
    uint32_t b = blockIdx.x;
    if (a.xcd_map) {
      const uint32_t xcd = b & 7u;
      uint32_t slot = b >> 3;
      if constexpr (HALF) {   // (both halves and every query tile of a chunk on one XCD: they share its L2)
        half = slot & 1u;
        slot >>= 1;
      }
      qt = slot % a.q_tiles;
      chunk = xcd * (a.n_chunks >> 3) + slot / a.q_tiles;
    } else {
      if constexpr (HALF) {
        half = b & 1u;
        b >>= 1;
      }
      qt = b % a.q_tiles;
      chunk = b / a.q_tiles;
    }
    half = (uint32_t)__builtin_amdgcn_readfirstlane((int)half);
    // (a division by a run-time value goes through the vector ALU: without this the tile range, the loop bound and the
    // DMA sources derived from it would live in vector registers — and, this kernel being out of them, in scratch)
    qt = (uint32_t)__builtin_amdgcn_readfirstlane((int)qt);
    chunk = (uint32_t)__builtin_amdgcn_readfirstlane((int)chunk);
  }
  // wave (wr, wc) takes rows [128 wr, +128) x queries [64 wc, +64) of the tile; HALF: the workgroup's four waves are the
  // four wc of row half wr = half
  const int wr = HALF ? (int)half : (w >> 2), wc = w & 3;
  const uint32_t lrow0 = HALF ? 0u : (uint32_t)wr * 128u;   // the wave's first row inside the X stage / row-parameter block in LDS
  float4* qp_lds = (float4*)(smem + L::kQpOff);
  float* qinv_lds = (float*)(smem + L::kQinvOff);

  if (tid < (int)kTileQ) {  // query parameters and this pass's threshold, once per workgroup
    const size_t qg = (size_t)qt * kTileQ + tid;
    float4 qp = a.qparams[qg];
    qp.w = a.thr ? a.thr[qg] : __builtin_inff();
    qp_lds[tid] = qp;
    // 1 / s_q, erring low (the alarm threshold K = num * this must err towards alarms); +inf: the query's step is 0,
    // S_lower does not depend on I and nothing can alarm
    qinv_lds[tid] = qp.x > 0.0f ? (1.0f - 1e-5f) / qp.x : __builtin_inff();
  }
  if (tid < 64) ((uint32_t*)(smem + L::kSyncOff))[tid] = 0u;
  if (tid < 8) ((uint32_t*)(smem + L::kStgCntOff))[tid] = 0u;
  if (tid == 0) {
    I8Ctx* ctx = (I8Ctx*)(smem + L::kCtxOff);
    ctx->pool = a.pool;
    ctx->pool_cnt = a.pool_cnt;
    ctx->ovf = a.ovf;
    ctx->perm = a.perm;
    ctx->rowp = a.rowp;
    ctx->pool_cap = a.pool_cap;
    ctx->n = a.n;
    ctx->q_tile0 = qt * kTileQ;
    ctx->pad = 0u;
  }

  const uint32_t tile_begin = a.tile0 + chunk * a.tiles_per_chunk;
  uint32_t tile_end = tile_begin + a.tiles_per_chunk;
  if (tile_end > a.tile0 + a.n_tiles) tile_end = a.tile0 + a.n_tiles;
  // (uniform by construction; said so, or the loop bound lives in a vector register — and, this kernel being out of
  // them, in scratch, reloaded behind an s_waitcnt vmcnt(0) once per tile)
  const uint32_t my_tiles =
      (uint32_t)__builtin_amdgcn_readfirstlane((int)(tile_end > tile_begin ? tile_end - tile_begin : 0u));
  const uint32_t ktiles = a.ld / kRowBI8;  // stages per tile (a.ld % 64 == 0)

  // ---- DMA duty of this wave: 1-KiB pieces w and w+8 of the X stage block and of the Q stage block (linear
  // copies: the blocks are stored in HBM in the LDS image, scan8_index); waves 0..3 also one piece each of the
  // tile's row parameters (256 x 16 B), once per tile.  Sources are uniform pointers advanced one block per stage.
  const uint32_t voff = (uint32_t)lane * 16u;
  const uint32_t voff8 = voff + L::kXStage / 2;   // the wave's second piece of a stage: 8 (HALF: 4) KiB further on
  const size_t tile_bytes = (size_t)ktiles * kStageI8;
  // (HALF: rows [128 half, +128) of a stage block are its bytes [8192 half, +8192) — the LDS image is row-major)
  const char* xsrc = (const char*)a.X + (size_t)tile_begin * tile_bytes + (size_t)half * L::kXStage * (HALF ? 1u : 0u) +
                     (size_t)w * 1024;
  const char* qbase = (const char*)a.Q + (size_t)qt * ((size_t)(ktiles + 3) * kStageI8) + (size_t)w * 1024;
  const char* qsrc = qbase;
  const uint32_t rpiece = (uint32_t)w & (L::kRowpWaves - 1u);
  const char* rsrc = (const char*)(a.rowp + (size_t)tile_begin * kTileRows16 + (HALF ? (size_t)half * 128u : 0u)) +
                     (size_t)rpiece * 1024;
  const uint32_t xdst = L::kXOff + (uint32_t)w * 1024u;  // + slot * kXStage (+ kXStage / 2 for the second piece)
  const uint32_t qdst = L::kQOff + (uint32_t)w * 1024u;
  const uint32_t rdst = L::kRowpOff + rpiece * 1024u;  // + (tile % 3) * kRowpSlot

#define EHX_DMA(DST_BASE, DST_IMM, VOFF, SRC)                                                              \
  do {                                                                                                     \
    asm volatile("s_add_u32 m0, %0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3"                       \
                 :                                                                                         \
                 : "s"(DST_BASE), "n"(DST_IMM), "v"(VOFF), "s"(SRC)                                        \
                 : "memory", "scc");                                                                       \
  } while (0)
  // (the ring slot of a stage is a run-time value in the second loop — a tile may be any number of stages long — so
  // the LDS destination comes in a scalar register)
#define EHX_DMA_RT(DST, VOFF, SRC)                                                       \
  do {                                                                                   \
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"         \
                 :                                                                       \
                 : "s"(DST), "v"(VOFF), "s"(SRC)                                         \
                 : "memory");                                                            \
  } while (0)
#define EHX_DMA_X0(SLOT) EHX_DMA(xdst, (SLOT) * L::kXStage, voff, xsrc)
#define EHX_DMA_Q0(SLOT) EHX_DMA(qdst, (SLOT) * 16384, voff, qsrc)
#define EHX_DMA_X1(SLOT) EHX_DMA(xdst, (SLOT) * L::kXStage + L::kXStage / 2, voff8, xsrc)
#define EHX_DMA_Q1(SLOT)                               \
  do {                                                 \
    EHX_DMA(qdst, (SLOT) * 16384 + 8192, voff8, qsrc); /* (full-tile kernel only: voff8 = voff + 8192) */ \
    xsrc += kStageI8;                                  \
    qsrc += kStageI8;                                  \
  } while (0)

  // ---- fragment read offsets: lane l reads the 16-byte chunk (l >> 4) of row / query (l & 15) of its block (any fixed
  // assignment of k to lanes is fine as long as rows and queries use the same one); the chunk's physical place is
  // swizzled by the row (scan8_swz) so that the four 16-lane groups a ds_read_b128 is served in hit 16 different slots
  const uint32_t sw = scan8_swz((uint32_t)j15);
  uint32_t a_off = L::kXOff + (lrow0 + (uint32_t)j15) * kRowBI8 + (((uint32_t)qd) ^ sw) * 16u;  // + rb*1024
  uint32_t b_off = L::kQOff + (uint32_t)(wc * 64 + j15) * kRowBI8 + (((uint32_t)qd) ^ sw) * 16u;   // + cb*1024
  // (opaque to the compiler: it would fold kQOffI8 = 64 KiB into every read's constant, find that the sum no longer
  // fits the instruction's 16-bit offset field, and keep one address register per (ring slot, block) — 24 registers
  // spilled to scratch, whose reloads sit in the stage loop behind s_waitcnt vmcnt(0).  As two opaque bases every
  // read is base + immediate.)
  asm volatile("" : "+v"(a_off), "+v"(b_off));

  i32x4 acc[8][4];
#pragma unroll
  for (int rb = 0; rb < 8; ++rb)
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[rb][cb][r] = 0;

  uint32_t rp_slot = 0u;  // LDS slot (tile % 3) of the current tile's row parameters
  uint32_t stg_n = 0u;    // keys in this wave's staging buffer (wave-uniform; the buffer is the wave's own)
  // tile parameters of the current tile: a scalar load issued a whole tile before its use (the epilogue must not
  // wait for a global round trip)
  // (through the CONSTANT address space: real scalar loads, counted by lgkmcnt.  As plain global pointers they compiled
  // to vector loads of a uniform address, and — the compiler cannot count the DMA pieces the inline asm issues — to an
  // s_waitcnt vmcnt(0) at every tile boundary: the whole look-ahead of the ring drained once per tile, rounds 2 and 3.)
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  typedef const __attribute__((address_space(4))) f32x4* cf4p;
  auto ldc = [](cf4p p, size_t i) -> float4 {
    const f32x4 v = p[i];
    return make_float4(v.x, v.y, v.z, v.w);
  };
  auto uniform_ptr = [](const void* p) -> cf4p {  // (the address IS uniform; this tells the compiler so)
    const uint64_t v = (uint64_t)p;
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return (cf4p)(((uint64_t)hi << 32) | lo);
  };
  // (a chunk past the end of the pass — chunks x tiles_per_chunk rounds up — owns no tile and must not touch memory beyond the
  // arrays' two padding entries: its look-ahead loads read the pass's first tile instead.  Round 6: the unclamped load of
  // tile_begin ran up to 15 KiB past a 50-KiB tileg array and faulted once the allocator placed it at the end of a block.)
  const uint32_t tile_par = my_tiles ? tile_begin : a.tile0;
  const cf4p tilep_c = uniform_ptr(a.tilep + tile_par);
  float4 tp_cur = ldc(tilep_c, 0);
  // max |A| of this wave's four 32-row lane groups (one per l >> 4) of the current tile: uniform, loaded a tile ahead
  const cf4p tgp = uniform_ptr(a.tileg + (size_t)tile_par * 16 + (size_t)wr * 4);
  float4 tg_cur = ldc(tgp, 0);
  // ... and the B margins of the same four groups (tileg[tile][8 + g]: min B of the group - min B of the tile)
  float4 tgb_cur = ldc(tgp, 2);
  // =============================== tile epilogue ===============================
  auto epilogue = [&](uint32_t t) {
    const uint32_t tile = tile_begin + t;
    // (HALF: "the tile" of this workgroup is rows [128 half, +128) of it — row parameters, positions and lane rows are
    // counted from there)
    const uint32_t tile_row0 = tile * kTileRows16 + (HALF ? half * 128u : 0u);
    const uint32_t rp_off = L::kRowpOff + rp_slot * L::kRowpSlot;  // this (half) tile's row parameters in LDS
    const float4* rp = (const float4*)(smem + rp_off);
    // (the lane's coordinates are derived afresh, behind an opaque copy of the lane id: left to itself the compiler
    // keeps a dozen epilogue addresses alive across the stage loop, which has no registers to spare)
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    const int j15 = lane_e & 15, qd = lane_e >> 4;
    const int col0 = wc * 64 + j15;                          // + 16 cb: this lane's four queries
    const uint32_t rbase = lrow0 + 4u * (uint32_t)qd;  // + 16 rb + r: this lane's 32 rows
    if (DUMP) {
      // Blocks of 16 queries x 16 rows, a query's 16 rows contiguous (scan8_dump_index, ehx_kernels.h): the four
      // accumulators of a block are four consecutive rows of one query — one 16-byte store — and the 64 lanes of a store
      // cover ONE contiguous KiB (lane (j15, qd): query j15 of the block, rows 4 qd .. 4 qd + 3); sample_select256_kernel
      // reads a query's scores as 64-byte runs.  (Round 6.  Row-major before: 4-byte stores, and the select read one element
      // per 4-KiB stride; plainly query-major — 16 queries 8 KiB apart per store — made this kernel 9 us slower.)
      const size_t n_s = (size_t)a.n_tiles * kTileRows16;
      float* const o0 = a.dump + scan8_dump_index((uint32_t)qt * kTileQ + (uint32_t)col0,
                                                  tile_row0 - a.tile0 * kTileRows16 + rbase, (uint32_t)n_s);
      const size_t cb_step = (n_s >> 4) << 8;   // the next block of 16 queries
      const float4 q0 = qp_lds[col0], q1 = qp_lds[col0 + 16], q2 = qp_lds[col0 + 32], q3 = qp_lds[col0 + 48];
#pragma unroll
      for (int rb = 0; rb < 8; ++rb) {
        float4 s0, s1, s2, s3;
        float* const f0 = (float*)&s0;
        float* const f1 = (float*)&s1;
        float* const f2 = (float*)&s2;
        float* const f3 = (float*)&s3;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float4 P = rp[rbase + 16u * (uint32_t)rb + (uint32_t)r];
          f0[r] = i8_score(P, q0, acc[rb][0][r]);
          f1[r] = i8_score(P, q1, acc[rb][1][r]);
          f2[r] = i8_score(P, q2, acc[rb][2][r]);
          f3[r] = i8_score(P, q3, acc[rb][3][r]);
        }
        float* const o = o0 + 256 * rb;   // the next 16 rows: the next KiB
        *(float4*)(o) = s0;
        *(float4*)(o + cb_step) = s1;
        *(float4*)(o + 2 * cb_step) = s2;
        *(float4*)(o + 3 * cb_step) = s3;
        asm volatile("" ::: "memory");  // (one row block at a time: 32 rows of parameters held at once spill)
      }
      return;
    }
#if EHX_I8_ABL & 32
    // (ablation: no epilogue arithmetic at all, the accumulators merely kept alive — without a use the compiler deletes the MFMAs)
#pragma unroll
    for (int rb = 0; rb < 8; ++rb)
      asm volatile("" : : "v"(acc[rb][0]), "v"(acc[rb][1]), "v"(acc[rb][2]), "v"(acc[rb][3]));
    return;
#endif
    const float4 tp = tp_cur;  // uniform: (max|A|, max|C|, max|D|, min B) of the tile's rows (loaded a tile ago)
    // max |A_r| over this lane's 32 rows (tiles ordered by step: every row of the group has exactly this |A_r|)
    const float4 tg = tg_cur;
    const float gm = qd == 0 ? tg.x : (qd == 1 ? tg.y : (qd == 2 ? tg.z : tg.w));
    // (gm = 0: a group of padding rows — +inf, nothing alarms unless K = -inf.  v_rcp_f32 is good to 1 ulp: three
    // roundings against a margin of 2e-6 — the level still errs low; the IEEE division was a dozen instructions per tile)
    const float rgm = (1.0f - 2e-6f) * __builtin_amdgcn_rcpf(gm);
    // The alarm level K of a (tile, query) is computed ONCE per wave — lane (qd, j15) takes query 16 qd + j15 of the
    // wave's 64 — and handed to the four lanes that hold the query's accumulators by a lane permute.  -inf, +inf or > 0.
    const float k_own = i8_alarm_k(tp, qp_lds[wc * 64 + lane_e], qinv_lds[wc * 64 + lane_e]);
    // L2^2 (B_r = |x_r|^2 varies inside a tile): every row of this lane's group has B_r >= min B of the tile + dB, so the
    // group's level is K + dB gamma_q / s_q — the factor per query, erring LOW (towards alarms), handed round like K.
    // (gamma_q < 0 never happens for L2^2; clamped so that a margin can only ever raise the level soundly.  -inf + inf or
    // 0 * inf give NaN, which the clamp below turns into "always alarm".)
    float g_own = 0.0f, dB = 0.0f;
    if (a.group_b) {
      const float4 tgb = tgb_cur;
      dB = qd == 0 ? tgb.x : (qd == 1 ? tgb.y : (qd == 2 ? tgb.z : tgb.w));
      g_own = fmaxf(qp_lds[wc * 64 + lane_e].z, 0.0f) * qinv_lds[wc * 64 + lane_e] * (1.0f - 1e-4f);
    }
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
      // ---- phase 1: can ANY of the lane's 32 accumulators of this query belong to a candidate?  I |A_r| >= K with
      // |A_r| = gm for the whole group (K > 0: a negative I never qualifies; K = -inf: always), i.e. I >= K / gm: ONE
      // integer level per lane and query (rounded down: a superset, the hit path judges every key exactly) against the
      // integer maximum ----
      float kq = __shfl(k_own, cb * 16 + j15, 64);
      if (a.group_b) kq = __builtin_fmaf(dB, __shfl(g_own, cb * 16 + j15, 64), kq);
      // (clamped before the conversion: -2.1e9 = always, 2.1e9 = never — |I| <= 2048 * 127^2)
      const int ti = (int)fminf(fmaxf(kq * rgm, -2.1e9f), 2.1e9f);
      int m8[8];
#pragma unroll
      for (int rb = 0; rb < 8; ++rb) {
        const i32x4 c = acc[rb][cb];
        m8[rb] = max(max(c[0], c[1]), max(c[2], c[3]));
      }
      const int im = max(max(max(m8[0], m8[1]), max(m8[2], m8[3])), max(max(m8[4], m8[5]), max(m8[6], m8[7])));
#if EHX_I8_ABL & 2
      if (im >= ti) asm volatile("" ::: "memory");  // (the comparison stays, the slow path does not)
      continue;
#endif
      EHX_CNT(0);
      if (!__any(im >= ti)) continue;
      EHX_CNT(1);
      // ---- phase 2: which accumulators?  First the row blocks whose own maximum reaches the level (the eight per-block
      // maxima are at hand), then their four accumulators.  A 32-bit mask per lane (bit 4 rb + r), then one set bit per
      // lane and trip. ----
      const int ql = cb * 16 + j15;
      uint32_t pend = 0u;
#pragma unroll
      for (int rb = 7; rb >= 0; --rb) {  // (nibble by nibble: the bit constants stay inline operands)
        uint32_t nib = 0u;
        if (__any(m8[rb] >= ti)) {
          EHX_CNT(2);
          const i32x4 c = acc[rb][cb];
#pragma unroll
          for (int r = 0; r < 4; ++r) nib |= (c[r] >= ti) ? (1u << r) : 0u;
        }
        pend = (pend << 4) | nib;
      }
      while (__any(pend != 0u)) {
        EHX_CNT(3);
        const bool hi = pend != 0u;
        int v = 0;
        uint32_t r_local = rbase;
        if (hi) {
          const int b = __builtin_ctz(pend);
          pend &= pend - 1u;
          i32x4 c4 = acc[0][cb];
#pragma unroll
          for (int rb = 1; rb < 8; ++rb) {
            const bool pick = (b >> 2) == rb;
            const i32x4 o = acc[rb][cb];
            c4[0] = pick ? o[0] : c4[0];
            c4[1] = pick ? o[1] : c4[1];
            c4[2] = pick ? o[2] : c4[2];
            c4[3] = pick ? o[3] : c4[3];
          }
          const int r = b & 3;
          v = r == 0 ? c4[0] : (r == 1 ? c4[1] : (r == 2 ? c4[2] : c4[3]));
          r_local = rbase + 16u * (uint32_t)(b >> 2) + (uint32_t)r;
        }
        i8_hit<HALF>(v, hi, r_local, tile_row0, ql, w, stg_n);
      }
    }
  };

  // ---- lock-step with the sibling workgroups of this chunk (see the header) ----
  // per-sibling progress words [chunk][4], checked once per tile (REV loop only; the launcher guarantees n_chunks * 4 words)
  uint32_t* const sync_ctr = (a.sync && a.sync_tol > 0u && a.xcd_map && a.q_tiles > 1 && REV && a.q_tiles <= 4)
                                 ? a.sync + chunk * 4u
                                 : nullptr;
  bool sync_on = sync_ctr != nullptr && !DUMP && !HALF;
  const uint32_t sync_m0 = L::kSyncOff;

  __syncthreads();  // state init visible
#if EHX_I8_PRIO
  // Static priority for the arbitration loser (MI355X_MICROARCH.md, "Two waves per SIMD" item 4): of the two waves a SIMD
  // holds, the younger one gets the leftover issue slots on every stage; ONE s_setprio 1 for it, no per-stage flips.
  {
    uint32_t hwid_p;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid_p));
    if (hwid_p & 1u) __builtin_amdgcn_s_setprio(1);   // (hardware wave slot of the SIMD: the second-resident wave)
  }
#endif
  if constexpr (HALF) {
    // Two half-tile workgroups share a CU so that one's tile epilogue (vector ALU) runs beside the other's matrix work —
    // but two identical workgroups launched together run IN PHASE: both in their matrix stages, then both in their
    // epilogues.  The second-resident wave of every SIMD (hardware wave slot, HW_ID[0]) therefore starts a.skew x 64
    // cycles late — about half a tile — and the two stay out of phase because they run at the same rate.
    // Measured (profiles/r05_r_skew.jsonl, 6.25 M x 128, scan phase per batch): skew 0 1.031-1.037 ms, 16 / 32 1.023,
    // 48 0.987-0.998, 64 0.985, 96 0.996; 1 M x 128: no difference (0.290).  Short chunks are not worth the wait.
    if (a.skew && my_tiles >= 16u) {
      uint32_t hwid;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
      if (hwid & 1u) {   // (s_sleep N = 64 N cycles; whole blocks of 32, then the remainder in steps of 1: EHX_I8_SKEW is exact)
        for (uint32_t i = 32; i <= a.skew; i += 32) __builtin_amdgcn_s_sleep(32);
        for (uint32_t i = 0; i < (a.skew & 31u); ++i) __builtin_amdgcn_s_sleep(1);
      }
    }
  }
  if (my_tiles > 0) {  // (a chunk past the end of the pass has nothing to scan and must not touch memory)
  // ---- prologue: row parameters of tile 0, stages 0..2 into ring slots 0..2 ----
  // (the row parameters travel through LDS for the sample pass only — DUMP, which scores every accumulator; a scan pass
  // stages its hits unjudged and the flush reads the parameters of those few rows from HBM: round 6)
  if (DUMP && w < (int)L::kRowpWaves) EHX_DMA(rdst, 0, voff, rsrc);
  // QRES (short rows: a tile is at most four stages, ld <= 256): the query tile's stage blocks — the same for every
  // row tile — are copied ONCE into the four slots of the Q ring and stay there; a stage then copies its two X pieces
  // per wave only (half the DMA instructions, half the L2 -> LDS bytes), and the query fragments of stage ks of a
  // tile are read from slot ks.  The counted wait of a stage is 4 instead of 8 (two pieces per stage and wave).
  constexpr int kStageWait = QRES ? 4 : 8;
  if constexpr (QRES) {
    for (uint32_t ks = 0; ks < ktiles; ++ks) {
      const uint32_t qd0 = qdst + (ks << 14);
      // the 16 1-KiB pieces of a query stage block: two per wave (pieces w, w + 8), HALF: four (w, w + 4, w + 8, w + 12)
      constexpr uint32_t kQStep = HALF ? 4096u : 8192u;
#pragma unroll
      for (uint32_t pc = 0; pc < 16u / L::kWaves; ++pc) {
        const uint32_t qdp = qd0 + pc * kQStep, vo = voff + pc * kQStep;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(qdp), "v"(vo), "s"(qsrc) : "memory");
      }
      qsrc += kStageI8;
    }
    EHX_DMA_X0(0); EHX_DMA_X1(0); xsrc += kStageI8;
    EHX_DMA_X0(1); EHX_DMA_X1(1); xsrc += kStageI8;
    EHX_DMA_X0(2); EHX_DMA_X1(2); xsrc += kStageI8;
  } else {
    EHX_DMA_X0(0); EHX_DMA_Q0(0); EHX_DMA_X1(0); EHX_DMA_Q1(0);
    EHX_DMA_X0(1); EHX_DMA_Q0(1); EHX_DMA_X1(1); EHX_DMA_Q1(1);
    EHX_DMA_X0(2); EHX_DMA_Q0(2); EHX_DMA_X1(2); EHX_DMA_Q1(2);
  }
  wait_vmcnt<kStageWait>();  // stage 0 (and what is older: row parameters, the resident query tile) landed
  lds_barrier_i8();  // B_0
#if EHX_I8_ABL & 4
  wait_vmcnt<0>();
#define EHX_SDMA(DST, VOFF, SRC) do { } while (0)
#else
#define EHX_SDMA(DST, VOFF, SRC) EHX_DMA_RT(DST, VOFF, SRC)
#endif

  // Fragments.  The eight row-block fragments of a stage are single-buffered: block rb's fragment is dead after its
  // four MFMAs and is re-read for the NEXT stage in place; the four query-block fragments feed every row block and
  // are double-buffered (fb0 / fb1 swap roles every stage).  All twelve reads of the next stage are issued in the
  // second half of a stage, after its barrier has made the next stage visible.
  i32x4 fa[8], fb0[4], fb1[4];
#pragma unroll
  for (int rb = 0; rb < 8; ++rb) fa[rb] = *(const i32x4*)(smem + a_off + rb * 1024);
#pragma unroll
  for (int cb = 0; cb < 4; ++cb) fb0[cb] = *(const i32x4*)(smem + b_off + cb * 1024);

#if EHX_I8_ABL & 8
#define EHX_FR(P) fa[0]
#else
#define EHX_FR(P) (*(const i32x4*)(P))
#endif
#define EHX_MF(B, RB, CB) acc[RB][CB] = EHX_MFMA_I8(fa[RB], B[CB], acc[RB][CB])
  // the first (for this shape: only the first) MFMA of a block in a tile's first stage starts from the constant 0 (an
  // inline operand) instead of 128 v_mov per wave after every epilogue
  const i32x4 zero4 = {0, 0, 0, 0};
#define EHX_MFZ(B, RB, CB) acc[RB][CB] = EHX_MFMA_I8(fa[RB], B[CB], zero4)
#define EHX_SB() __builtin_amdgcn_sched_barrier(0)
#if EHX_I8_ABL & 16
#define EHX_STAGE_BARRIER() do { } while (0)
#else
#define EHX_STAGE_BARRIER() lds_barrier_i8()
#endif
#if EHX_I8_ABL & 4
#define EHX_SDMA_X0(S) do { } while (0)
#define EHX_SDMA_Q0(S) do { } while (0)
#define EHX_SDMA_X1(S) do { } while (0)
#define EHX_SDMA_Q1(S) do { } while (0)
#else
#define EHX_SDMA_X0(S) EHX_DMA_X0(S)
#define EHX_SDMA_Q0(S) EHX_DMA_Q0(S)
#define EHX_SDMA_X1(S) EHX_DMA_X1(S)
#define EHX_SDMA_Q1(S) EHX_DMA_Q1(S)
#endif
  // One stage = 32 MFMAs, every accumulator block once.  First half: row blocks 0-3 (fragments loaded during the
  // previous stage) and this wave's four DMA pieces of the stage three ahead (its ring slot was last read two stages
  // ago); then the stage barrier (own pieces counted: the two younger stages may still be in flight); second half:
  // row blocks 4-7 and the twelve fragment reads of the next stage — one other instruction at most between two MFMAs.
#define EHX_STAGE16_BODY(M0, BC, BN, AN, BNX, DX0, DQ0, DX1, DQ1)                                         \
  do {                                                                                                   \
    M0(BC, 0, 0); DX0;                                     EHX_SB();                                     \
    M0(BC, 0, 1);                                          EHX_SB();                                     \
    M0(BC, 0, 2); DQ0;                                     EHX_SB();                                     \
    M0(BC, 0, 3);                                          EHX_SB();                                     \
    M0(BC, 1, 0); DX1;                                     EHX_SB();                                     \
    M0(BC, 1, 1);                                          EHX_SB();                                     \
    M0(BC, 1, 2); DQ1;                                     EHX_SB();                                     \
    M0(BC, 1, 3);                                          EHX_SB();                                     \
    M0(BC, 2, 0);                                          EHX_SB();                                     \
    M0(BC, 2, 1);                                          EHX_SB();                                     \
    M0(BC, 2, 2);                                          EHX_SB();                                     \
    M0(BC, 2, 3);                                          EHX_SB();                                     \
    M0(BC, 3, 0);                                          EHX_SB();                                     \
    M0(BC, 3, 1);                                          EHX_SB();                                     \
    M0(BC, 3, 2);                                          EHX_SB();                                     \
    M0(BC, 3, 3);                                          EHX_SB();                                     \
    /* stage barrier: the next stage landed and is visible; nobody reads the slot the stage after the    \
       next two will be copied into */                                                                   \
    wait_vmcnt<kStageWait>();                                                                            \
    EHX_STAGE_BARRIER();                                                                                 \
    EHX_SB();                                                                                            \
    M0(BC, 4, 0); BN[0] = EHX_FR(smem + (BNX));           EHX_SB();                                     \
    M0(BC, 4, 1); BN[1] = EHX_FR(smem + (BNX) + 1024);    EHX_SB();                                     \
    M0(BC, 4, 2); BN[2] = EHX_FR(smem + (BNX) + 2048);    EHX_SB();                                     \
    M0(BC, 4, 3); BN[3] = EHX_FR(smem + (BNX) + 3072);    EHX_SB();                                     \
    M0(BC, 5, 0); fa[0] = EHX_FR(smem + (AN));            EHX_SB();                                     \
    M0(BC, 5, 1); fa[1] = EHX_FR(smem + (AN) + 1024);     EHX_SB();                                     \
    M0(BC, 5, 2); fa[2] = EHX_FR(smem + (AN) + 2048);     EHX_SB();                                     \
    M0(BC, 5, 3); fa[3] = EHX_FR(smem + (AN) + 3072);     EHX_SB();                                     \
    M0(BC, 6, 0); fa[4] = EHX_FR(smem + (AN) + 4096);     EHX_SB();                                     \
    M0(BC, 6, 1); fa[5] = EHX_FR(smem + (AN) + 5120);     EHX_SB();                                     \
    M0(BC, 6, 2);                                          EHX_SB();                                     \
    M0(BC, 6, 3);                                          EHX_SB();                                     \
    M0(BC, 7, 0); fa[6] = EHX_FR(smem + (AN) + 6144);     EHX_SB();                                     \
    M0(BC, 7, 1);                                          EHX_SB();                                     \
    M0(BC, 7, 2);                                          EHX_SB();                                     \
    M0(BC, 7, 3);                                          EHX_SB();                                     \
    fa[7] = EHX_FR(smem + (AN) + 7168);                    EHX_SB();                                     \
  } while (0)
  // ring slot S a compile-time constant: no address arithmetic at all
#define EHX_STAGE16_CT(S, M0, BC, BN)                                                                    \
  do {                                                                                                   \
    constexpr uint32_t sn = (uint32_t)(((S) + 1) & 3) * kStageI8;                                        \
    constexpr int sd = ((S) + 3) & 3;                                                                    \
    EHX_STAGE16_BODY(M0, BC, BN, a_off + sn, b_off + sn, EHX_SDMA_X0(sd), EHX_SDMA_Q0(sd), EHX_SDMA_X1(sd), \
                     EHX_SDMA_Q1(sd));                                                                   \
  } while (0)

  // Two loops over the same stage body.  REV (a tile is a whole number of ring revolutions, ld % 256 == 0: d = 768, 1536,
  // 1024, 512, 256 ...): the ring slot of every stage is a compile-time constant.  Otherwise (d = 128, 384, 640 ...):
  // one loop over single stages whose slot is a scalar.
  if constexpr (REV) {
    // Tile by tile; a tile is `kquads` revolutions of the ring (ld % 256 == 0).  The first stage of a tile is its own
    // copy of the stage body: its MFMAs start the accumulators from 0.
    const uint32_t kquads = ktiles >> 2;
    // row parameters of tile 1 (consumed by its epilogue, a whole tile from now)
    rsrc += kTileRows16 * 16;
    if (DUMP && w < (int)L::kRowpWaves) EHX_DMA(rdst, L::kRowpSlot, voff, rsrc);
    // Lock-step by TILE (round 5; a.sync_tol = the tolerance in tiles): each of the chunk's query-tile workgroups
    // publishes how many tiles it has completed in a word of its own (a plain store: no read-modify-write, no return
    // value to wait for) and, once per tile, looks at a snapshot of its siblings' words taken a tile earlier (an
    // asynchronous global->LDS load: stale by a tile, and progress only grows, so a stale view shows less than the truth
    // and at worst sends the wave to the live poll, which decides).  Only a workgroup that really is more than sync_tol
    // tiles ahead of the slowest sibling waits.  What the four stream stays inside the XCD's L2: 8 chunks x
    // (sync_tol + 1) tiles x 192 KiB at d = 768.  One foreign store per tile in the vmcnt queue (see the header: more
    // than one outstanding could let a counted wait pass early; a tile is ~10 us, the store retires in ~1).
    auto after_tile = [&](const uint32_t t) {
      if (sync_on && w == 0) {
        // (the snapshot is read with a REAL LDS instruction: through the volatile generic pointer the compiler emitted
        // flat_load_dword + s_waitcnt vmcnt(0) — the whole DMA look-ahead of the wave drained at every look, which is what
        // the lock-step "cost" in rounds 2-4 and in this round's first measurement: +7.7 % by tile, +60 % by revolution)
        i32x4 snap;
        {
          const uint32_t snap_at = L::kSyncOff;
          asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(snap) : "v"(snap_at) : "memory");
        }
        uint32_t mn = 0xFFFFFFFFu;
        for (uint32_t i = 0; i < a.q_tiles; ++i) mn = min(mn, (uint32_t)__builtin_amdgcn_readfirstlane(snap[i]));
        const uint32_t need = t + 1u > a.sync_tol ? t + 1u - a.sync_tol : 0u;
        if (mn < need) {
          uint32_t spins = 0;
          for (;;) {
            mn = 0xFFFFFFFFu;
            for (uint32_t i = 0; i < a.q_tiles; ++i)
              mn = min(mn, (uint32_t)__builtin_amdgcn_readfirstlane(
                               __hip_atomic_load(sync_ctr + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)));
            if (mn >= need) break;
            __builtin_amdgcn_s_sleep(8);
            if (++spins > 4000u) {  // a sibling is not resident (or died): never hang, just stop synchronising
              sync_on = false;
              break;
            }
          }
        }
        if (lane == 0) __hip_atomic_store(sync_ctr + qt, t + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // (the lane id behind an opaque copy: a loop-invariant derived from it would be kept alive across the stage
        // loop, which has no register to spare)
        int lane_s = lane;
        asm volatile("" : "+v"(lane_s));
        const uint32_t voff4 = ((uint32_t)lane_s & 3u) * 4u;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2 sc1"
                     :
                     : "s"(sync_m0), "v"(voff4), "s"(sync_ctr)
                     : "memory");
      }
    };
    for (uint32_t t = 0; t < my_tiles; ++t) {
      EHX_STAGE16_CT(0, EHX_MFZ, fb0, fb1);
      EHX_STAGE16_CT(1, EHX_MF, fb1, fb0);
      EHX_STAGE16_CT(2, EHX_MF, fb0, fb1);
      EHX_STAGE16_CT(3, EHX_MF, fb1, fb0);
      for (uint32_t kq = 1; kq < kquads; ++kq) {
        EHX_STAGE16_CT(0, EHX_MF, fb0, fb1);
        EHX_STAGE16_CT(1, EHX_MF, fb1, fb0);
        EHX_STAGE16_CT(2, EHX_MF, fb0, fb1);
        EHX_STAGE16_CT(3, EHX_MF, fb1, fb0);
      }
      // (the lock-step's store and snapshot load go out BEFORE the epilogue: the rows of tile t are consumed — that is
      // what the siblings' L2 window is about — and the store, which retires out of order and makes the next counted wait
      // one entry stricter while it is in flight, gets the epilogue and half a stage to come back: issued after the
      // epilogue it cost 5 % of the scan time, profiles/r05_l_sync.jsonl)
      after_tile(t);
#if !(EHX_I8_ABL & 1)
      epilogue(t);
#endif
      // next tile: its stage 3 is the next one to issue (stages 0..2 came from the repeated blocks); row
      // parameters of the tile after it (past the last tile: the array's two tiles of tail padding)
      qsrc = qbase + 3 * kStageI8;
      rsrc += kTileRows16 * 16;
      tp_cur = ldc(tilep_c, t + 1);  // (past the last tile: the array's padding entries)
      tg_cur = ldc(tgp, (size_t)(t + 1) * 4);
      tgb_cur = ldc(tgp, (size_t)(t + 1) * 4 + 2);
      // tile t+2 goes to the slot tile t-1 used: every wave left that epilogue long ago
      const uint32_t rp_next = rp_slot == 0u ? 2u : rp_slot - 1u;  // (t + 2) % 3 == (t - 1) % 3
      rp_slot = rp_slot == 2u ? 0u : rp_slot + 1u;
      if (DUMP && w < (int)L::kRowpWaves) {
        const uint32_t rd = rdst + rp_next * L::kRowpSlot;
        EHX_DMA(rd, 0, voff, rsrc);
      }
    }
  } else {
    // One flat loop over the stages of the chunk; a tile is `ktiles` of them (ld / 64: any number).  The tile boundary
    // work hangs off a counter and may fall anywhere in a revolution: nothing in the ring depends on where a tile starts
    // (X stages stream linearly, Q stages repeat with period ktiles, and the three blocks appended to the Q array cover
    // the look-ahead across the boundary).  The query fragments change hands by a copy here (a tile may be an odd
    // number of stages).
    const uint32_t total_stages = my_tiles * ktiles;
    // row parameters of tile 1 (consumed by its epilogue, a whole tile from now)
    rsrc += kTileRows16 * 16;
    if (DUMP && w < (int)L::kRowpWaves) EHX_DMA(rdst, L::kRowpSlot, voff, rsrc);
    uint32_t ks = 0, t = 0, slot = 0;
    // one stage whose ring slot is a run-time value: STAGE is the stage body to use (first stage of a tile or not)
#define EHX_RT_DX0 EHX_SDMA(dx0, voff, xsrc)
#define EHX_RT_DQ0 do { if constexpr (!QRES) EHX_SDMA(dq0, voff, qsrc); } while (0)
#define EHX_RT_DX1 EHX_SDMA(dx0 + L::kXStage / 2, voff8, xsrc)
#define EHX_RT_DQ1 do { if constexpr (!QRES) EHX_SDMA(dq0 + 8192u, voff8, qsrc); } while (0)
  // (QRES: the next stage's query fragments come from the resident slot of ITS index inside its tile)
#define EHX_RT_STAGE(STAGE)                                                                  \
  do {                                                                                       \
    const uint32_t sn = ((slot + 1u) & 3u) << L::kXShift, sd = ((slot + 3u) & 3u) << L::kXShift; \
    const uint32_t dx0 = xdst + sd, dq0 = qdst + sd;                                         \
    const uint32_t ksn_ = ks + 1u == ktiles ? 0u : ks + 1u;                                  \
    const uint32_t an_ = a_off + sn, bn_ = b_off + (QRES ? (ksn_ << 14) : sn);               \
    (void)dq0;                                                                               \
    STAGE;                                                                                   \
    xsrc += kStageI8;                                                                        \
    if constexpr (!QRES) qsrc += kStageI8;                                                   \
    _Pragma("unroll") for (int cb = 0; cb < 4; ++cb) fb0[cb] = fb1[cb];                      \
    slot = (slot + 1u) & 3u;                                                                 \
    ++st;                                                                                    \
  } while (0)
    auto tile_done = [&]() {  // t: the tile that has just been completed
#if !(EHX_I8_ABL & 1)
      epilogue(t);
#endif
      ++t;
      // next tile: its stage 3 is the next one to issue (stages 0..2 came from the repeated blocks); row
      // parameters of the tile after it (past the last tile: the array's two tiles of tail padding)
      qsrc = qbase + 3 * kStageI8;
      rsrc += kTileRows16 * 16;
      tp_cur = ldc(tilep_c, t);  // (past the last tile: the array's padding entries)
      tg_cur = ldc(tgp, (size_t)t * 4);
      tgb_cur = ldc(tgp, (size_t)t * 4 + 2);
      // tile t+1 (counting the new t) goes to the slot tile t-2 used: every wave left that epilogue long ago
      const uint32_t rp_next = rp_slot == 0u ? 2u : rp_slot - 1u;  // (t + 1) % 3 == (t - 2) % 3
      rp_slot = rp_slot == 2u ? 0u : rp_slot + 1u;
      if (DUMP && w < (int)L::kRowpWaves) {
        const uint32_t rd = rdst + rp_next * L::kRowpSlot;
        EHX_DMA(rd, 0, voff, rsrc);
      }
    };
    uint32_t st = 0;
    if constexpr (HALF) {
      // Rows of 128 bytes (d = 128, the shape HALF exists for): a tile is exactly TWO stages, two tiles are one revolution of
      // the ring — every slot a compile-time constant again, the query fragments change hands by name instead of sixteen
      // register copies per stage (behind an lgkmcnt(0): the next stage's fragment reads had to LAND before the stage could
      // end), and the per-stage slot arithmetic is gone.  Same stage body, same order of MFMAs, DMA pieces and fragment
      // reads as the run-time-slot loop below: bit-identical.  (Round 6.)
      if (ktiles == 2u) {
#define EHX_NODMA do { } while (0)
#define EHX_TILE2_CT(S)                                                                                               \
  do {                                                                                                                \
    EHX_STAGE16_BODY(EHX_MFZ, fb0, fb1, a_off + (uint32_t)(((S) + 1) & 3) * L::kXStage, b_off + (1u << 14),           \
                     EHX_SDMA_X0(((S) + 3) & 3), EHX_NODMA, EHX_SDMA_X1(((S) + 3) & 3), EHX_NODMA);                   \
    xsrc += kStageI8;                                                                                                 \
    EHX_STAGE16_BODY(EHX_MF, fb1, fb0, a_off + (uint32_t)(((S) + 2) & 3) * L::kXStage, b_off,                         \
                     EHX_SDMA_X0(((S) + 4) & 3), EHX_NODMA, EHX_SDMA_X1(((S) + 4) & 3), EHX_NODMA);                   \
    xsrc += kStageI8;                                                                                                 \
    tile_done();                                                                                                      \
  } while (0)
        uint32_t left = my_tiles;
#pragma unroll 1
        for (; left >= 2u; left -= 2u) {
          EHX_TILE2_CT(0);
          EHX_TILE2_CT(2);
        }
        if (left) EHX_TILE2_CT(0);   // (a third copy of the tile, run once per chunk; folding it into the loop with two exits spilled 164 registers)
#undef EHX_TILE2_CT
#undef EHX_NODMA
        st = total_stages;
      }
    }
    if constexpr (PAIR) {
      // (an instantiation of its own: inside the one-stage kernel the second loop cost the first four spilled registers, one of
      // them reloaded behind an s_waitcnt vmcnt(0) in every stage)
      // Rows of an EVEN number of stages that is no multiple of four (d = 384, 640, 896 ...: ld % 128 == 0): the stages go in
      // pairs — the query fragments change hands by name (fb0 -> fb1 -> fb0), no sixteen register copies per stage behind an
      // lgkmcnt(0), as in the two loops above; a tile boundary always falls between two pairs.  Slots stay run-time values.
      // (Round 6, after the two-stage loop of the half-tile kernel: -10 % there.)
      {
#define EHX_RT_STAGE2(M0, BC, BN)                                                               \
  do {                                                                                          \
    const uint32_t sn = ((slot + 1u) & 3u) << L::kXShift, sd = ((slot + 3u) & 3u) << L::kXShift; \
    const uint32_t dx0 = xdst + sd, dq0 = qdst + sd;                                            \
    const uint32_t an_ = a_off + sn, bn_ = b_off + sn;                                          \
    EHX_STAGE16_BODY(M0, BC, BN, an_, bn_, EHX_RT_DX0, EHX_RT_DQ0, EHX_RT_DX1, EHX_RT_DQ1);     \
    xsrc += kStageI8;                                                                           \
    qsrc += kStageI8;                                                                           \
    slot = (slot + 1u) & 3u;                                                                    \
    ++st;                                                                                       \
  } while (0)
#pragma unroll 1
        while (st < total_stages) {   // (total_stages is even)
          if (ks == 0u) EHX_RT_STAGE2(EHX_MFZ, fb0, fb1);
          else EHX_RT_STAGE2(EHX_MF, fb0, fb1);
          EHX_RT_STAGE2(EHX_MF, fb1, fb0);
          ks += 2u;
          if (ks == ktiles) {
            ks = 0;
            tile_done();
          }
        }
#undef EHX_RT_STAGE2
      }
    }
    if constexpr (!PAIR) {
#pragma unroll 1
      while (st < total_stages) {
        if (ks == 0u)
          EHX_RT_STAGE(EHX_STAGE16_BODY(EHX_MFZ, fb0, fb1, an_, bn_, EHX_RT_DX0, EHX_RT_DQ0, EHX_RT_DX1, EHX_RT_DQ1));
        else
          EHX_RT_STAGE(EHX_STAGE16_BODY(EHX_MF, fb0, fb1, an_, bn_, EHX_RT_DX0, EHX_RT_DQ0, EHX_RT_DX1, EHX_RT_DQ1));
        if (++ks == ktiles) {
          ks = 0;
          tile_done();
        }
      }
    }
#undef EHX_RT_STAGE
#undef EHX_RT_DX0
#undef EHX_RT_DQ0
#undef EHX_RT_DX1
#undef EHX_RT_DQ1
  }
  }  // my_tiles > 0
#undef EHX_MF
#undef EHX_SB
#undef EHX_FR
#undef EHX_SDMA
#undef EHX_DMA_RT
#undef EHX_SDMA_X0
#undef EHX_SDMA_Q0
#undef EHX_SDMA_X1
#undef EHX_SDMA_Q1
#undef EHX_STAGE16_CT
#undef EHX_STAGE16_BODY
#undef EHX_STAGE_BARRIER
#undef EHX_MFZ
#undef EHX_DMA_X0
#undef EHX_DMA_Q0
#undef EHX_DMA_X1
#undef EHX_DMA_Q1
#undef EHX_DMA

  // ---- final: what is left in this wave's staging buffer goes to the pools ----
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  if (DUMP) return;
  if (lane == 0) ((uint32_t*)(smem + L::kStgCntOff))[w] = stg_n;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  i8_flush_staging<HALF>();
}

hipError_t launch_flat_scan_i8(const ScanArgsI8& a, hipStream_t st) {
  static DynLdsAttr attr;
  const void* fns[] = {(const void*)flat_scan_i8_kernel<false, true>, (const void*)flat_scan_i8_kernel<false, false>,
                       (const void*)flat_scan_i8_kernel<true, true>, (const void*)flat_scan_i8_kernel<true, false>,
                       (const void*)flat_scan_i8_kernel<false, false, true>,
                       (const void*)flat_scan_i8_kernel<false, false, true, true>,
                       (const void*)flat_scan_i8_kernel<false, false, false, false, true>};
  if (hipError_t e = attr.ensure(fns, (int)(sizeof(fns) / sizeof(fns[0])), I8L<false>::kLdsBytes); e != hipSuccess) return e;
  if (a.ld == 0 || a.ld % kRowBI8) return hipErrorInvalidValue;
  const uint32_t grid = a.q_tiles * a.n_chunks;
  const bool rev = a.ld % (4 * kRowBI8) == 0;  // whole ring revolutions per tile: the compile-time-slot loop
#define EHX_LAUNCH_I8(D, R) \
  hipLaunchKernelGGL((flat_scan_i8_kernel<D, R>), dim3(grid), dim3(I8L<false>::kThreads), I8L<false>::kLdsBytes, st, a)
  // short rows (a tile of at most four stages): the query tile resident in LDS (QRES in the kernel); EHX_I8_QRES=0: off
  const bool qres_on = env().i8_qres;
  if (a.dump) {
    if (rev) EHX_LAUNCH_I8(true, true);
    else EHX_LAUNCH_I8(true, false);
  } else {
    // short rows of at most two stages: two half-tile workgroups per CU (HALF in the kernel); EHX_I8_HALF=0: off
    const bool half_on = env().i8_half;
    if (rev) EHX_LAUNCH_I8(false, true);
    else if (qres_on && half_on && a.ld <= 2 * kRowBI8)
      hipLaunchKernelGGL((flat_scan_i8_kernel<false, false, true, true>), dim3(2 * grid), dim3(I8L<true>::kThreads),
                         I8L<true>::kLdsBytes, st, a);
    else if (qres_on && a.ld <= 4 * kRowBI8)
      hipLaunchKernelGGL((flat_scan_i8_kernel<false, false, true>), dim3(grid), dim3(I8L<false>::kThreads), I8L<false>::kLdsBytes, st, a);
    else if ((a.ld / kRowBI8) % 2u == 0u)   // an even number of stages per tile (d = 384, 640, 896 ...): stages in pairs
      hipLaunchKernelGGL((flat_scan_i8_kernel<false, false, false, false, true>), dim3(grid), dim3(I8L<false>::kThreads),
                         I8L<false>::kLdsBytes, st, a);
    else EHX_LAUNCH_I8(false, false);
  }
#undef EHX_LAUNCH_I8
  return hipGetLastError();
}

}  // namespace ehx
