// Microbenchmark: which WAVE TILE should the int8 scan's main loop use?  (round 3; follows i8_stage_sync.hip, whose
// result was: barriers and epilogue staggering are worth nothing, the loop's own shape — 8 waves, wave tile 128 x 64,
// 12 ds_read_b128 per 16 MFMAs — tops out at 57-59 % of the 5 POP/s int8 peak even without HBM traffic.)
// Per CU and 64-byte stage that shape reads 96 KB of fragments from LDS (768 cycles at 128 B/clk) next to 1024 cycles
// of v_mfma_i32_32x32x32_i8, and the direct-to-LDS fill of the stage writes another 32 KB: the LDS port is as busy as
// the matrix pipe.  Variants, all LDS-resident (no HBM), same operand data, 12 stages per tile:
//   regs    8 waves, MFMAs on register-resident fragments only (what the pipe sustains at this power state)
//   w128x64 8 waves (2 per SIMD), wave tile 128 rows x 64 queries: the shipped shape (12 reads / 16 MFMAs)
//   w128x128 4 waves (1 per SIMD, 512 registers), wave tile 128 x 128: 16 reads / 32 MFMAs = 2/3 of the LDS bytes per op
//   w256x64  4 waves, wave tile 256 rows x 64 queries: 20 reads / 32 MFMAs
// each with an epilogue of E rounds (0 / 1) of (convert, multiply, max) over the accumulators per tile.  The effective
// shader clock of every run is reported too (s_memtime ticks per 100-MHz wall tick).
//   hipcc --offload-arch=gfx950 -O3 -o i8_tile_shapes.bin scripts/ubench/i8_tile_shapes.hip && ./i8_tile_shapes.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int kStageBytes = 16384;  // 256 rows x 64 B
constexpr int kRing = 4;

// RB x CB blocks of 32 x 32 per wave; WAVES waves per workgroup; REGS: fragments never re-read from LDS
template <int WAVES, int RB, int CB, int E, bool REGS>
__global__ __launch_bounds__(WAVES * 64) void loop_kernel(const int4* __restrict__ src, float* out, long long* clk,
                                                          int tiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  constexpr int WC = 256 / (CB * 32);  // waves side by side over the 256 queries of the workgroup tile
  const int wr = w / WC, wc = w % WC;
  for (int i = tid; i < 2 * kRing * kStageBytes / 16; i += WAVES * 64) ((int4*)smem)[i] = src[i & 4095];
  __syncthreads();
  i32x16 acc[RB][CB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[rb][cb][r] = 0;
  const int i31 = lane & 31, h = lane >> 5;
  const unsigned sw = ((unsigned)i31 >> 2) & 3u;
  const unsigned a_row = (unsigned)((wr * RB * 32) % 256 + i31) * 64u;
  const unsigned b_row = (unsigned)(kRing * kStageBytes) + (unsigned)(wc * CB * 32 + i31) * 64u;
  float sink = 0.0f;
  const int stages = tiles * 12;
  long long t0 = 0, w0 = 0;
  if (blockIdx.x == 0 && tid == 0) {
    t0 = (long long)__builtin_readcyclecounter();
    w0 = (long long)wall_clock64();
  }
  i32x4 fa[RB], fb[CB];
  if (REGS) {
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) fa[rb] = *(const i32x4*)(smem + a_row + rb * 2048);
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) fb[cb] = *(const i32x4*)(smem + b_row + cb * 2048);
  }
  for (int s = 0; s < stages; ++s) {
    __syncthreads();
    const unsigned slot = (unsigned)(s % kRing) * kStageBytes;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      if (!REGS) {
        const unsigned chunk = (((unsigned)(2 * kk + h)) ^ sw) * 16u;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) fa[rb] = *(const i32x4*)(smem + slot + a_row + rb * 2048 + chunk);
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) fb[cb] = *(const i32x4*)(smem + slot + b_row + cb * 2048 + chunk);
      }
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
          acc[rb][cb] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[rb], fb[cb], acc[rb][cb], 0, 0, 0);
    }
    if (E > 0 && (s % 12) == 11) {
#pragma unroll
      for (int e = 0; e < E; ++e) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
          for (int cb = 0; cb < CB; ++cb) {
            float m = -1e30f;
#pragma unroll
            for (int r = 0; r < 16; ++r) m = fmaxf(m, (float)acc[rb][cb][r] * (1.0f + 0.001f * (float)(r + e)));
            sink = fmaxf(sink, m);
          }
      }
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[rb][cb][r] = 0;
    }
  }
  if (blockIdx.x == 0 && tid == 0) {
    clk[0] = (long long)__builtin_readcyclecounter() - t0;
    clk[1] = (long long)wall_clock64() - w0;
  }
  float t = sink;
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
      for (int r = 0; r < 16; ++r) t += (float)acc[rb][cb][r];
  out[blockIdx.x * (WAVES * 64) + tid] = t;
}

template <int WAVES, int RB, int CB, int E, bool REGS>
static void run(const char* name, const int4* d, float* out, long long* clk, int tiles) {
  const size_t lds = 2 * kRing * kStageBytes;
  auto k = loop_kernel<WAVES, RB, CB, E, REGS>;
  (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipFuncAttributes fa;
  (void)hipFuncGetAttributes(&fa, (const void*)k);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  k<<<256, WAVES * 64, lds>>>(d, out, clk, 8);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  k<<<256, WAVES * 64, lds>>>(d, out, clk, tiles);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  long long h[2] = {0, 1};
  (void)hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
  const double ops = 256.0 * tiles * 12 * WAVES * (RB * CB * 2) * 65536.0;
  printf("%-9s epilogue x%d: %8.3f ms  %7.1f TOP/s (%4.1f %% of 5000)  regs %3d  spill %d B  clock %.2f GHz\n", name, E, ms,
         ops / (ms * 1e-3) / 1e12, ops / (ms * 1e-3) / 1e12 / 50.0, fa.numRegs, (int)fa.localSizeBytes,
         h[1] ? (double)h[0] / (double)h[1] * 0.1 : 0.0);
}

int main() {
  std::vector<signed char> hbuf(65536);
  srand(3);
  for (auto& v : hbuf) {
    float s = 0;
    for (int j = 0; j < 12; ++j) s += (float)rand() / (float)RAND_MAX;
    float g = (s - 6.0f) * 40.0f;
    v = (signed char)(g > 127 ? 127 : (g < -127 ? -127 : g));
  }
  int4* d;
  float* out;
  long long* clk;
  (void)hipMalloc(&d, hbuf.size());
  (void)hipMemcpy(d, hbuf.data(), hbuf.size(), hipMemcpyHostToDevice);
  (void)hipMalloc(&out, 256 * 512 * 4);
  (void)hipMalloc(&clk, 16);
  const int tiles = 4000;
  run<8, 4, 2, 0, true>("regs", d, out, clk, tiles);
  run<8, 4, 2, 0, false>("w128x64", d, out, clk, tiles);
  run<8, 4, 2, 1, false>("w128x64", d, out, clk, tiles);
  run<4, 4, 4, 0, false>("w128x128", d, out, clk, tiles / 2 * 2);
  run<4, 4, 4, 1, false>("w128x128", d, out, clk, tiles);
  run<4, 8, 2, 0, false>("w256x64", d, out, clk, tiles);
  run<4, 8, 2, 1, false>("w256x64", d, out, clk, tiles);
  run<4, 4, 4, 0, true>("regs4w", d, out, clk, tiles);
  return 0;
}
