// Microbenchmark (round 6): short rows (a tile = TS stages of 64 bytes, TS = 2 for d = 128) — is the matrix pipe fed better
// by FOUR waves per SIMD with 64 x 64 wave tiles than by TWO with 128 x 64?  A lone wave issues v_mfma_i32_16x16x64_i8 every
// ~30 cycles, two waves of a SIMD together every ~17 (profiles/r04_a_i8_mfma_shapes_ubench.txt); the shipped short-row kernel
// runs two waves per SIMD, so whenever one of them is in its tile epilogue (vector ALU: an integer maximum over the lane's
// accumulators against a level, per query block) the other feeds the pipe alone, at half rate.
//   w128x64  8 waves per CU (2 per SIMD), wave tile 128 rows x 64 queries: 32 MFMAs + 12 ds_read_b128 per stage
//   w64x64  16 waves per CU (4 per SIMD), wave tile  64 rows x 64 queries: 16 MFMAs +  8 ds_read_b128 per stage
// Fragments re-read from LDS every stage (static data, no DMA), one barrier per stage, first stage of a tile starts its
// accumulators from 0 through the MFMA's C operand (no v_mov), epilogue E = 0 / 1 per tile: max3 tree + compare per query
// block, as in k_flati8.hip's phase 1.  Reports TOP/s and the effective shader clock.
//   hipcc --offload-arch=gfx950 -O3 -o i8_short_rows.bin scripts/ubench/i8_short_rows.hip && ./i8_short_rows.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int kStageBytes = 16384;  // 256 rows x 64 B
constexpr int kRing = 4;

template <int WAVES, int RB, int TS, int E>
__global__ __launch_bounds__(WAVES * 64) void loop_kernel(const int4* __restrict__ src, int* out, long long* clk, int tiles,
                                                          int level) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wr = w >> 2, wc = w & 3;   // WAVES / 4 row groups of RB * 16 rows, four query groups of 64
  for (int i = tid; i < 2 * kRing * kStageBytes / 16; i += WAVES * 64) ((int4*)smem)[i] = src[i & 4095];
  __syncthreads();
  int sink = 0;
  long long t0 = 0, w0 = 0;
  if (blockIdx.x == 0 && tid == 0) {
    t0 = (long long)__builtin_readcyclecounter();
    w0 = (long long)wall_clock64();
  }
  i32x4 acc[RB][4];
  const int i15 = lane & 15, q = lane >> 4;
  const unsigned sw = ((unsigned)i15 >> 2) & 3u;
  const unsigned a_row = (unsigned)(wr * RB * 16 + i15) * 64u + (((unsigned)q) ^ sw) * 16u;
  const unsigned b_row = (unsigned)(kRing * kStageBytes) + (unsigned)(wc * 64 + i15) * 64u + (((unsigned)q) ^ sw) * 16u;
  const i32x4 zero4 = {0, 0, 0, 0};
  i32x4 fa[RB], fb[4];
  int s = 0;
  for (int t = 0; t < tiles; ++t) {
#pragma unroll
    for (int ks = 0; ks < TS; ++ks, ++s) {
      __syncthreads();
      const unsigned slot = (unsigned)(s % kRing) * kStageBytes;
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) fa[rb] = *(const i32x4*)(smem + slot + a_row + rb * 1024);
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) fb[cb] = *(const i32x4*)(smem + slot + b_row + cb * 1024);
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
          acc[rb][cb] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[rb], fb[cb], ks == 0 ? zero4 : acc[rb][cb], 0, 0, 0);
    }
    if (E > 0) {
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) {
        int m8[RB];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
          const i32x4 c = acc[rb][cb];
          m8[rb] = max(max(c[0], c[1]), max(c[2], c[3]));
        }
        int im = m8[0];
#pragma unroll
        for (int rb = 1; rb < RB; ++rb) im = max(im, m8[rb]);
        const int ti = level + (lane & 3) + cb;   // (per-lane level: a convert and a multiply in the real kernel)
        if (__any(im >= ti)) sink += im;          // never taken (level is huge): the branch and the ballot stay
      }
    } else {
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)
        asm volatile("" : : "v"(acc[rb][0]), "v"(acc[rb][1]), "v"(acc[rb][2]), "v"(acc[rb][3]));
    }
  }
  if (blockIdx.x == 0 && tid == 0) {
    clk[0] = (long long)__builtin_readcyclecounter() - t0;
    clk[1] = (long long)wall_clock64() - w0;
  }
  out[blockIdx.x * (WAVES * 64) + tid] = sink;
}

template <int WAVES, int RB, int TS, int E>
static void run(const char* name, const int4* d, int* out, long long* clk, int tiles) {
  const size_t lds = 2 * kRing * kStageBytes;
  auto k = loop_kernel<WAVES, RB, TS, E>;
  (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipFuncAttributes fa;
  (void)hipFuncGetAttributes(&fa, (const void*)k);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  k<<<256, WAVES * 64, lds>>>(d, out, clk, 64, 1 << 30);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  k<<<256, WAVES * 64, lds>>>(d, out, clk, tiles, 1 << 30);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  long long h[2] = {0, 1};
  (void)hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
  const double ops = 256.0 * tiles * TS * 256.0 * 256.0 * 64.0 * 2.0;   // a CU's tile: 256 rows x 256 queries x 64 k per stage
  printf("%-10s %2d waves  %d stages per tile  epilogue x%d: %8.3f ms  %7.1f TOP/s (%4.1f %% of 5000)  regs %3d  spill %d B  clock %.2f GHz\n",
         name, WAVES, TS, E, ms, ops / (ms * 1e-3) / 1e12, ops / (ms * 1e-3) / 1e12 / 50.0, fa.numRegs, (int)fa.localSizeBytes,
         h[1] ? (double)h[0] / (double)h[1] * 0.1 : 0.0);
}

int main() {
  std::vector<signed char> hbuf(65536);
  srand(3);
  for (auto& b : hbuf) b = (signed char)(rand() % 255 - 127);
  int4* d;
  int* out;
  long long* clk;
  (void)hipMalloc(&d, 65536);
  (void)hipMalloc(&out, 256 * 1024 * 4);
  (void)hipMalloc(&clk, 16);
  (void)hipMemcpy(d, hbuf.data(), 65536, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 2; ++rep) {
    run<8, 8, 2, 0>("w128x64", d, out, clk, 20000);
    run<16, 4, 2, 0>("w64x64", d, out, clk, 20000);
    run<8, 8, 2, 1>("w128x64", d, out, clk, 20000);
    run<16, 4, 2, 1>("w64x64", d, out, clk, 20000);
    run<8, 8, 12, 1>("w128x64", d, out, clk, 4000);
    run<16, 4, 12, 1>("w64x64", d, out, clk, 4000);
  }
  return 0;
}
