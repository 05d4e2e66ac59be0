// Exhaustive ("flat") kNN on the gfx950 matrix cores.
//
// Replaces, for the brute-force configuration, the distance loop the reference runs inside
// hnswlib (searchKnn -> fstdistfunc_, call site embeddinghub/embeddingstore/index.cc:41): the
// B x N query-by-row distance matrix is a dense contraction, so it runs on
// v_mfma_f32_32x32x2_f32 (exact fp32, k-ordered fma chain) and is never materialised: each
// 128(rows) x 256(queries) tile is filtered in registers against per-query running thresholds and
// only the survivors reach a per-query candidate list.  A canonical re-rank (rerank_kernel below)
// then recomputes the surviving distances in exactly the oracle's (hnswlib SSE) summation order,
// so ids and distances are bit-identical to the exhaustive oracle.
//
// Layout / mapping (gfx950, wave64):
//   * workgroup = 256 threads = 4 waves, one per SIMD, 1 workgroup per CU; wave (wr, wc) =
//     (w>>1, w&1) owns 64 rows x 128 queries = 2x4 MFMA 32x32 blocks = 128 accumulator registers
//     (AGPRs), leaving the whole architectural VGPR file to fragments and the epilogue.  fp32 MFMA
//     issues every 64 cycles with 64-cycle dependent latency, so 8 independent accumulators from
//     one wave keep the SIMD's matrix pipe saturated;
//   * MFMA A = corpus rows, B = queries, so a lane's 16 accumulator values of one block all belong
//     to ONE query (col = lane&31) -> one threshold per block;
//   * both operand tiles ([128|256][32] fp32) are staged by global_load_lds (16 B/lane, no VGPR
//     round trip) into a double buffer; the 16-B chunk index of a row is XOR-swizzled with
//     (row>>1)&7 on the SOURCE address and on the ds_read_b128 side (LDS image stays lane-linear),
//     which makes the fragment reads bank-conflict free;
//   * a lane's ds_read_b128 gives 4 consecutive k for its (row, k-half); MFMA step t of a group
//     uses component t of both operands, i.e. the k order is permuted identically for A and B;
//   * grid = q_tiles x n_chunks persistent workgroups; each walks a contiguous range of row tiles
//     for one query tile, keeping per-query state (count, threshold key) in LDS and the 64
//     candidate slots per query in a per-block global scratch (L2 resident).  With grid % 8 == 0
//     the q_tiles blocks that stream the same rows are placed on the same XCD so the rows are
//     fetched from HBM once and hit in that XCD's L2 for the other query tiles.
#include <cstdlib>

#include "ehx_kernels.h"

namespace ehx {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int kThreads = 256;
constexpr uint32_t kXStage = kTileRows * kBK * 4;                  // 16 KiB
constexpr uint32_t kQStage = kTileQ * kBK * 4;                     // 32 KiB
constexpr uint32_t kXOff = 0;                                      // Xs[2]
constexpr uint32_t kQOff = 2 * kXStage;                            // Qs[2]
constexpr uint32_t kThrKeyOff = kQOff + 2 * kQStage;               // u64 thr_key[256]
constexpr uint32_t kThrFOff = kThrKeyOff + 256 * 8;                // f32 thr_f[256]
constexpr uint32_t kCntOff = kThrFOff + 256 * 4;                   // i32 cnt[256]
constexpr uint32_t kFlagOff = kCntOff + 256 * 4;                   // i32 flags[4]
constexpr uint32_t kLdsBytes = kFlagOff + 16;
static_assert(kTileRows == 128 && kTileQ == 256 && kBK == 32, "kernel geometry is hard-wired");

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst_uniform) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_dst_uniform, 16, 0, 0);
}

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

#define EHX_MFMA(A, B, C) __builtin_amdgcn_mfma_f32_32x32x2f32((A), (B), (C), 0, 0, 0)

}  // namespace

size_t scan_lds_bytes() { return kLdsBytes; }

__global__ __launch_bounds__(kThreads, 1) void flat_scan_kernel(const ScanArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = w >> 1, wc = w & 1;
  const int h = lane >> 5, i31 = lane & 31;

  // ---- block -> (query tile, chunk) ----
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
  uint64_t* thr_key = (uint64_t*)(smem + kThrKeyOff);
  float* thr_f = (float*)(smem + kThrFOff);
  int* cnt = (int*)(smem + kCntOff);
  int* flags = (int*)(smem + kFlagOff);
  uint64_t* cand = a.cand + (size_t)blockIdx.x * (256u * kCandSlots);

  thr_key[tid] = kKeyInf;
  thr_f[tid] = __builtin_inff();
  cnt[tid] = 0;
  if (tid < 4) flags[tid] = 0;

  const uint32_t tile_begin = a.tile0 + chunk * a.tiles_per_chunk;
  uint32_t tile_end = tile_begin + a.tiles_per_chunk;
  if (tile_end > a.tile0 + a.n_tiles) tile_end = a.tile0 + a.n_tiles;
  const uint32_t my_tiles = tile_end > tile_begin ? tile_end - tile_begin : 0u;
  const uint32_t ktiles = a.ld / kBK;
  const uint32_t total_steps = my_tiles * ktiles;

  // ---- per-lane constants for the staging loads ----
  // One global_load_lds instruction moves 8 tile rows x 128 B: lane L -> row ins*8+(L>>3), physical
  // 16-B chunk p = L&7, logical chunk c = p ^ ((row>>1)&7).  Per stage a wave issues 4 X
  // instructions (ins = w*4+u) and 8 Q instructions (ins = w*8+u).
  const float* Qtile = a.Q + (size_t)qt * kTileQ * a.ld;
  uint32_t stx_off[4], stq_off[8];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const uint32_t row = ((uint32_t)w * 4 + u) * 8 + (lane >> 3);
    stx_off[u] = row * a.ld + (((lane & 7) ^ ((row >> 1) & 7)) * 4);
  }
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const uint32_t row = ((uint32_t)w * 8 + u) * 8 + (lane >> 3);
    stq_off[u] = row * a.ld + (((lane & 7) ^ ((row >> 1) & 7)) * 4);
  }

  // ---- per-lane constants for the fragment reads ----
  // row r = base + i31 ; chunk for group j = (2j+h) ^ ((r>>1)&7) = (2j) ^ hs, hs = h ^ ((i31>>1)&7)
  const uint32_t hs = (uint32_t)h ^ ((uint32_t)(i31 >> 1) & 7u);
  uint32_t joff[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) joff[j] = (((uint32_t)(2 * j)) ^ hs) * 16;
  const uint32_t a_row_off = (uint32_t)(wr * 64 + i31) * 128;   // + rb*4096
  const uint32_t b_row_off = (uint32_t)(wc * 128 + i31) * 128;  // + cb*4096

  f32x16 acc[2][4];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[rb][cb][r] = 0.0f;

  // ---- software-pipelined main loop -------------------------------------------------------
  // A stage (one BK=32 slice of both operand tiles) is consumed as 4 groups of 32 MFMAs; the
  // fragments of group g+1 are read from LDS into the idle register set while group g's MFMAs
  // issue, and the single workgroup barrier per stage plus the staging loads of stage s+2 sit
  // inside stage s's last group, so the matrix pipe does not wait for LDS latency or DMA issue.
  f32x4 fa0[2], fb0[4], fa1[2], fb1[4];

#define EHX_GROUP(A, B, An, Bn, XS, QS)                                   \
  do {                                                                    \
    An[0] = *(const f32x4*)((XS));                                        \
    An[1] = *(const f32x4*)((XS) + 4096);                                 \
    _Pragma("unroll") for (int cb = 0; cb < 4; ++cb) {                   \
      acc[0][cb] = EHX_MFMA(A[0][0], B[cb][0], acc[0][cb]);               \
      acc[1][cb] = EHX_MFMA(A[1][0], B[cb][0], acc[1][cb]);               \
    }                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                    \
    Bn[0] = *(const f32x4*)((QS));                                        \
    Bn[1] = *(const f32x4*)((QS) + 4096);                                 \
    _Pragma("unroll") for (int cb = 0; cb < 4; ++cb) {                   \
      acc[0][cb] = EHX_MFMA(A[0][1], B[cb][1], acc[0][cb]);               \
      acc[1][cb] = EHX_MFMA(A[1][1], B[cb][1], acc[1][cb]);               \
    }                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                    \
    Bn[2] = *(const f32x4*)((QS) + 2 * 4096);                             \
    Bn[3] = *(const f32x4*)((QS) + 3 * 4096);                             \
    _Pragma("unroll") for (int cb = 0; cb < 4; ++cb) {                   \
      acc[0][cb] = EHX_MFMA(A[0][2], B[cb][2], acc[0][cb]);               \
      acc[1][cb] = EHX_MFMA(A[1][2], B[cb][2], acc[1][cb]);               \
    }                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                    \
    _Pragma("unroll") for (int cb = 0; cb < 4; ++cb) {                   \
      acc[0][cb] = EHX_MFMA(A[0][3], B[cb][3], acc[0][cb]);               \
      acc[1][cb] = EHX_MFMA(A[1][3], B[cb][3], acc[1][cb]);               \
    }                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                    \
  } while (0)

  if (total_steps > 0) {
    {  // stage 0 -> buffer 0
      const float* Xt = (const float*)a.X + (size_t)tile_begin * kTileRows * a.ld;
      char* gxs = smem + kXOff + (uint32_t)w * 4096;
      char* gqs = smem + kQOff + (uint32_t)w * 8192;
#pragma unroll
      for (int u = 0; u < 4; ++u) glds16(Xt + stx_off[u], gxs + u * 1024);
#pragma unroll
      for (int u = 0; u < 8; ++u) glds16(Qtile + stq_off[u], gqs + u * 1024);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    fa0[0] = *(const f32x4*)(smem + kXOff + a_row_off + joff[0]);
    fa0[1] = *(const f32x4*)(smem + kXOff + a_row_off + 4096 + joff[0]);
#pragma unroll
    for (int c = 0; c < 4; ++c) fb0[c] = *(const f32x4*)(smem + kQOff + b_row_off + c * 4096 + joff[0]);
    if (total_steps > 1) {  // stage 1 -> buffer 1
      const uint32_t nt = 1u / ktiles, nkt = 1u - nt * ktiles;
      const float* Xt = (const float*)a.X + (size_t)(tile_begin + nt) * kTileRows * a.ld + nkt * kBK;
      const float* Qt = Qtile + nkt * kBK;
      char* gxs = smem + kXOff + kXStage + (uint32_t)w * 4096;
      char* gqs = smem + kQOff + kQStage + (uint32_t)w * 8192;
#pragma unroll
      for (int u = 0; u < 4; ++u) glds16(Xt + stx_off[u], gxs + u * 1024);
#pragma unroll
      for (int u = 0; u < 8; ++u) glds16(Qt + stq_off[u], gqs + u * 1024);
    }
  }

  uint32_t kt = 0, t = 0;
  for (uint32_t step = 0; step < total_steps; ++step) {
    const uint32_t buf = step & 1;
    const char* xs = smem + kXOff + buf * kXStage + a_row_off;
    const char* qs = smem + kQOff + buf * kQStage + b_row_off;
    const char* xn = smem + kXOff + (buf ^ 1) * kXStage + a_row_off;
    const char* qn = smem + kQOff + (buf ^ 1) * kQStage + b_row_off;
    EHX_GROUP(fa0, fb0, fa1, fb1, xs + joff[1], qs + joff[1]);
    EHX_GROUP(fa1, fb1, fa0, fb0, xs + joff[2], qs + joff[2]);
    EHX_GROUP(fa0, fb0, fa1, fb1, xs + joff[3], qs + joff[3]);
    // ---- last group of the stage (fragments in set 1) with the hand-over inside ----
    const bool has_next = step + 1 < total_steps;
    const bool has_next2 = step + 2 < total_steps;
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
      acc[0][cb] = EHX_MFMA(fa1[0][0], fb1[cb][0], acc[0][cb]);
      acc[1][cb] = EHX_MFMA(fa1[1][0], fb1[cb][0], acc[1][cb]);
    }
    __builtin_amdgcn_sched_barrier(0);
    // stage s+1 has landed (issued a whole stage ago) and every wave has finished reading stage s
    // (its last fragments were fetched during group 2): one barrier, then refill buffer `buf`.
    if (has_next) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      fa0[0] = *(const f32x4*)(xn + joff[0]);
      fa0[1] = *(const f32x4*)(xn + 4096 + joff[0]);
#pragma unroll
      for (int c = 0; c < 4; ++c) fb0[c] = *(const f32x4*)(qn + c * 4096 + joff[0]);
    }
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
      acc[0][cb] = EHX_MFMA(fa1[0][1], fb1[cb][1], acc[0][cb]);
      acc[1][cb] = EHX_MFMA(fa1[1][1], fb1[cb][1], acc[1][cb]);
    }
    __builtin_amdgcn_sched_barrier(0);
    const uint32_t nstep = step + 2;
    const uint32_t nt = has_next2 ? nstep / ktiles : 0u;
    const uint32_t nkt = has_next2 ? nstep - nt * ktiles : 0u;
    const float* Xt = (const float*)a.X + (size_t)(tile_begin + nt) * kTileRows * a.ld + nkt * kBK;
    const float* Qt = Qtile + nkt * kBK;
    char* gxs = smem + kXOff + buf * kXStage + (uint32_t)w * 4096;
    char* gqs = smem + kQOff + buf * kQStage + (uint32_t)w * 8192;
    if (has_next2) {
#pragma unroll
      for (int u = 0; u < 2; ++u) glds16(Xt + stx_off[u], gxs + u * 1024);
#pragma unroll
      for (int u = 0; u < 4; ++u) glds16(Qt + stq_off[u], gqs + u * 1024);
    }
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
      acc[0][cb] = EHX_MFMA(fa1[0][2], fb1[cb][2], acc[0][cb]);
      acc[1][cb] = EHX_MFMA(fa1[1][2], fb1[cb][2], acc[1][cb]);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (has_next2) {
#pragma unroll
      for (int u = 2; u < 4; ++u) glds16(Xt + stx_off[u], gxs + u * 1024);
#pragma unroll
      for (int u = 4; u < 8; ++u) glds16(Qt + stq_off[u], gqs + u * 1024);
    }
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
      acc[0][cb] = EHX_MFMA(fa1[0][3], fb1[cb][3], acc[0][cb]);
      acc[1][cb] = EHX_MFMA(fa1[1][3], fb1[cb][3], acc[1][cb]);
    }
    __builtin_amdgcn_sched_barrier(0);

    if (++kt == ktiles) {
      kt = 0;
      // ================= tile epilogue: threshold filter + candidate append =================
      // Phase 1 (branch-free, fully unrolled): approximate distance s = dot*a_row + b_row of every
      // accumulator, recorded only as one bit "s <= threshold of its query" (4 words x 32 bits per
      // lane; word = rb*2 + (cb>>1), bit = (cb&1)*16 + reg).  The accumulators are left untouched.
      // Phase 2 (rare, loops): only lanes with set bits extract the dot product with a select
      // chain, recompute s with the same fma, and append (score,id) keys to the candidate slots.
      const uint32_t tile_row0 = (tile_begin + t) * kTileRows;
      const int qbase = wc * 128 + i31;
      uint32_t pend[4] = {0u, 0u, 0u, 0u};
      {
        float thrf[4];
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) thrf[cb] = thr_f[qbase + cb * 32];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
#pragma unroll
          for (int reg = 0; reg < 16; ++reg) {
            const uint32_t r = (uint32_t)(wr * 64 + rb * 32 + (reg & 3) + 8 * (reg >> 2)) + 4u * h;
            const float2 ab = a.rowp[tile_row0 + r];
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
              const float sc = __builtin_fmaf(acc[rb][cb][reg], ab.x, ab.y);
              pend[rb * 2 + (cb >> 1)] |= (sc <= thrf[cb]) ? (1u << ((cb & 1) * 16 + reg)) : 0u;
            }
          }
        }
      }
      const bool lane_any = (pend[0] | pend[1] | pend[2] | pend[3]) != 0u;
      // block-uniform decision (LDS flag) so every wave takes the same barrier path
      if (lane_any) flags[1] = 1;
      __syncthreads();
      const int tile_hot = flags[1];
      __syncthreads();
      if (tile_hot) {
        if (tid == 0) flags[1] = 0;
        // kprime < kCandSlots guarantees progress (a compacted list has free slots); the round
        // bound only turns a logic error into an error code instead of a hung GPU
        for (int round = 0;; ++round) {
#pragma unroll
          for (int wd = 0; wd < 4; ++wd) {
            const int rb = wd >> 1, cp = wd & 1;
            uint32_t retry = 0u;
            while (__any(pend[wd] != 0u)) {
              if (pend[wd] != 0u) {
                const int b = __builtin_ctz(pend[wd]);
                pend[wd] &= pend[wd] - 1u;
                float dot = 0.0f;
#pragma unroll
                for (int i = 0; i < 16; ++i) dot = (b == i) ? acc[rb][2 * cp][i] : dot;
#pragma unroll
                for (int i = 0; i < 16; ++i) dot = (b == 16 + i) ? acc[rb][2 * cp + 1][i] : dot;
                const int reg = b & 15;
                const int cb = 2 * cp + (b >> 4);
                const uint32_t r = (uint32_t)(wr * 64 + rb * 32 + (reg & 3) + 8 * (reg >> 2)) + 4u * h;
                const uint32_t grow = tile_row0 + r;
                const float2 ab = a.rowp[grow];
                const float sc = __builtin_fmaf(dot, ab.x, ab.y);
                const int q = qbase + cb * 32;
                const uint64_t key = ((uint64_t)f32_to_ordered(sc) << 32) | grow;
                if (grow < a.n && key < thr_key[q]) {
                  const int pos = atomicAdd(&cnt[q], 1);
                  if (pos < (int)kCandSlots) {
                    cand[q * kCandSlots + pos] = key;
                  } else {
                    flags[0] = 1;
                    retry |= 1u << b;
                  }
                }
              }
            }
            pend[wd] = retry;
          }
          __syncthreads();
          // ---- compaction: wave w owns queries w*64 .. w*64+63 ----
          const int overflow = flags[0];
          const int trigger = (int)a.kprime + ((int)kCandSlots - (int)a.kprime) / 2;
          {
            const int c = cnt[w * 64 + lane];
            const bool need = c >= trigger || (overflow && c > (int)kCandSlots);
            uint64_t mask = __ballot(need);
            while (mask) {
              const int qq = __builtin_ctzll(mask);
              mask &= mask - 1;
              const int q = w * 64 + qq;
              const int cq = cnt[q];
              const int nv = cq < (int)kCandSlots ? cq : (int)kCandSlots;
              uint64_t key = lane < nv ? cand[q * kCandSlots + lane] : kKeyInf;
              key = wave_sort64(key, lane);
              if (lane < (int)a.kprime) cand[q * kCandSlots + lane] = key;
              const uint64_t kth = __shfl(key, (int)a.kprime - 1, 64);
              if (lane == 0) {
                cnt[q] = nv < (int)a.kprime ? nv : (int)a.kprime;
                if (nv >= (int)a.kprime) {
                  thr_key[q] = kth;
                  thr_f[q] = ordered_to_f32((uint32_t)(kth >> 32));
                }
              }
            }
          }
          __syncthreads();
          if (!overflow) break;
          if (round >= 512) {
            if (tid == 0) atomicAdd(a.err, 1u);
            break;
          }
          if (tid == 0) flags[0] = 0;
          __syncthreads();
        }
      }
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[rb][cb][r] = 0.0f;
      ++t;
    }
  }
#undef EHX_GROUP

  // ---- final: sort every query's slots and publish the top-k' keys of this chunk ----
  __syncthreads();
  for (int qq = 0; qq < 64; ++qq) {
    const int q = w * 64 + qq;
    const int cq = cnt[q];
    const int nv = cq < (int)kCandSlots ? cq : (int)kCandSlots;
    uint64_t key = lane < nv ? cand[q * kCandSlots + lane] : kKeyInf;
    key = wave_sort64(key, lane);
    if (lane < (int)a.kprime)
      a.part[((size_t)(qt * kTileQ + q) * a.lists_total + a.list0 + chunk) * a.kprime + lane] = key;
  }
}

static int scan_variant() {
  static const int variant = [] {
    const char* v = getenv("EHX_SCAN_VARIANT");
    return v ? atoi(v) : 8;
  }();
  return variant;
}
uint32_t scan_lists_per_chunk() { return scan_variant() == 4 ? 1u : 2u; }

hipError_t launch_flat_scan(const ScanArgs& a, hipStream_t st) {
  const int variant = scan_variant();
  if (variant == 4 && a.x_half) return hipErrorInvalidValue;  // the 4-wave A/B variant scans fp32 rows only
  return variant == 4 ? launch_flat_scan4(a, st) : launch_flat_scan8(a, st);
}

hipError_t launch_flat_scan4(const ScanArgs& a, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)flat_scan_kernel,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  const uint32_t grid = a.q_tiles * a.n_chunks;
  hipLaunchKernelGGL(flat_scan_kernel, dim3(grid), dim3(kThreads), kLdsBytes, st, a);
  return hipGetLastError();
}

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

hipError_t launch_merge_lists(const uint64_t* ids, const float* dist, const uint32_t* count, uint32_t nq,
                              uint32_t k, uint32_t n_lists, uint64_t* out_ids, float* out_dist,
                              uint32_t* out_count, hipStream_t st, size_t ids_stride, size_t dist_stride,
                              size_t count_stride, uint64_t id_mul, uint64_t id_step) {
  hipLaunchKernelGGL(merge_lists_kernel, dim3(nq), dim3(64), 0, st, ids, dist, count, nq, k, n_lists, ids_stride,
                     dist_stride, count_stride, id_mul, id_step, out_ids, out_dist, out_count);
  return hipGetLastError();
}

}  // namespace ehx
