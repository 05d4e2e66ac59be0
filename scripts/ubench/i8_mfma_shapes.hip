// Microbenchmark (round 4): does v_mfma_i32_16x16x64_i8 sustain more than v_mfma_i32_32x32x32_i8?
// The guide lists ">= 3944 TOPS (16x16x64)" as the measured int8 ceiling; round 3's yardstick — the register-resident
// 32x32x32 loop of i8_tile_shapes.hip — held 3350 (8 waves per CU) / 3626 (4 waves) at 1.68 / 1.83 GHz.  Same harness,
// same operand data, same wave tile (128 rows x 64 queries = 128 accumulator registers, 12 ds_read_b128 per 64-byte
// stage), both instruction shapes side by side:
//   regs     MFMAs on register-resident fragments only
//   lds      fragments re-read from LDS every k-step (the shipped loop's traffic: 12 reads per 1 Mi int-ops)
// each with E rounds (0 / 1) of a per-accumulator epilogue per 12-stage tile; 8 waves (two per SIMD) and 4 waves.
// The effective shader clock of every run is reported (s_memtime ticks per 100-MHz wall tick).
//   hipcc --offload-arch=gfx950 -O3 -o i8_mfma_shapes.bin scripts/ubench/i8_mfma_shapes.hip && ./i8_mfma_shapes.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int kStageBytes = 16384;  // 256 rows x 64 B
constexpr int kRing = 4;

// SHAPE 32: 4 x 2 blocks of 32x32 (k = 32 per MFMA, two k-steps per stage); SHAPE 16: 8 x 4 blocks of 16x16 (k = 64)
template <int WAVES, int SHAPE, int E, bool REGS>
__global__ __launch_bounds__(WAVES * 64) void loop_kernel(const int4* __restrict__ src, float* out, long long* clk,
                                                          int tiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wr = (w >> 2) & 1, wc = w & 3;
  for (int i = tid; i < 2 * kRing * kStageBytes / 16; i += WAVES * 64) ((int4*)smem)[i] = src[i & 4095];
  __syncthreads();
  float sink = 0.0f;
  const int stages = tiles * 12;
  long long t0 = 0, w0 = 0;
  if (blockIdx.x == 0 && tid == 0) {
    t0 = (long long)__builtin_readcyclecounter();
    w0 = (long long)wall_clock64();
  }
  if constexpr (SHAPE == 32) {
    i32x16 acc[4][2];
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[rb][cb][r] = 0;
    const int i31 = lane & 31, h = lane >> 5;
    const unsigned sw = ((unsigned)i31 >> 2) & 3u;
    const unsigned a_row = (unsigned)(wr * 128 + i31) * 64u;
    const unsigned b_row = (unsigned)(kRing * kStageBytes) + (unsigned)(wc * 64 + i31) * 64u;
    i32x4 fa[4], fb[2];
    if (REGS) {
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) fa[rb] = *(const i32x4*)(smem + a_row + rb * 2048);
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) fb[cb] = *(const i32x4*)(smem + b_row + cb * 2048);
    }
    for (int s = 0; s < stages; ++s) {
      __syncthreads();
      const unsigned slot = (unsigned)(s % kRing) * kStageBytes;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        if (!REGS) {
          const unsigned chunk = (((unsigned)(2 * kk + h)) ^ sw) * 16u;
#pragma unroll
          for (int rb = 0; rb < 4; ++rb) fa[rb] = *(const i32x4*)(smem + slot + a_row + rb * 2048 + chunk);
#pragma unroll
          for (int cb = 0; cb < 2; ++cb) fb[cb] = *(const i32x4*)(smem + slot + b_row + cb * 2048 + chunk);
        }
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
#pragma unroll
          for (int cb = 0; cb < 2; ++cb)
            acc[rb][cb] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[rb], fb[cb], acc[rb][cb], 0, 0, 0);
      }
      if (E > 0 && (s % 12) == 11) {
#pragma unroll
        for (int e = 0; e < E; ++e)
#pragma unroll
          for (int rb = 0; rb < 4; ++rb)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
              float m = -1e30f;
#pragma unroll
              for (int r = 0; r < 16; ++r) m = fmaxf(m, (float)acc[rb][cb][r] * (1.0f + 0.001f * (float)(r + e)));
              sink = fmaxf(sink, m);
            }
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
#pragma unroll
          for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[rb][cb][r] = 0;
      }
    }
    float t = sink;
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) t += (float)acc[rb][cb][r];
    sink = t;
  } else {
    i32x4 acc[8][4];
#pragma unroll
    for (int rb = 0; rb < 8; ++rb)
#pragma unroll
      for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[rb][cb][r] = 0;
    // A fragment of a 16 x 64 block: lane l holds 16 bytes of row l % 16, k-chunk l / 16 (one ds_read_b128)
    const int i15 = lane & 15, q = lane >> 4;
    const unsigned sw = ((unsigned)i15 >> 2) & 3u;
    const unsigned a_row = (unsigned)(wr * 128 + i15) * 64u + (((unsigned)q) ^ sw) * 16u;
    const unsigned b_row = (unsigned)(kRing * kStageBytes) + (unsigned)(wc * 64 + i15) * 64u + (((unsigned)q) ^ sw) * 16u;
    i32x4 fa[8], fb[4];
    if (REGS) {
#pragma unroll
      for (int rb = 0; rb < 8; ++rb) fa[rb] = *(const i32x4*)(smem + a_row + rb * 1024);
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) fb[cb] = *(const i32x4*)(smem + b_row + cb * 1024);
    }
    for (int s = 0; s < stages; ++s) {
      __syncthreads();
      const unsigned slot = (unsigned)(s % kRing) * kStageBytes;
      if (!REGS) {
#pragma unroll
        for (int rb = 0; rb < 8; ++rb) fa[rb] = *(const i32x4*)(smem + slot + a_row + rb * 1024);
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) fb[cb] = *(const i32x4*)(smem + slot + b_row + cb * 1024);
      }
#pragma unroll
      for (int rb = 0; rb < 8; ++rb)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
          acc[rb][cb] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[rb], fb[cb], acc[rb][cb], 0, 0, 0);
      if (E > 0 && (s % 12) == 11) {
#pragma unroll
        for (int e = 0; e < E; ++e)
#pragma unroll
          for (int rb = 0; rb < 8; ++rb)
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
              float m = -1e30f;
#pragma unroll
              for (int r = 0; r < 4; ++r) m = fmaxf(m, (float)acc[rb][cb][r] * (1.0f + 0.001f * (float)(r + e)));
              sink = fmaxf(sink, m);
            }
#pragma unroll
        for (int rb = 0; rb < 8; ++rb)
#pragma unroll
          for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[rb][cb][r] = 0;
      }
    }
    float t = sink;
#pragma unroll
    for (int rb = 0; rb < 8; ++rb)
#pragma unroll
      for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r) t += (float)acc[rb][cb][r];
    sink = t;
  }
  if (blockIdx.x == 0 && tid == 0) {
    clk[0] = (long long)__builtin_readcyclecounter() - t0;
    clk[1] = (long long)wall_clock64() - w0;
  }
  out[blockIdx.x * (WAVES * 64) + tid] = sink;
}

template <int WAVES, int SHAPE, int E, bool REGS>
static void run(const char* name, const int4* d, float* out, long long* clk, int tiles) {
  const size_t lds = 2 * kRing * kStageBytes;
  auto k = loop_kernel<WAVES, SHAPE, E, REGS>;
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
  const double ops = 256.0 * tiles * 12 * WAVES * 16 * 65536.0;  // 1 Mi int-ops per wave and stage, either shape
  printf("%-22s %d waves  epilogue x%d: %8.3f ms  %7.1f TOP/s (%4.1f %% of 5000)  regs %3d  spill %d B  clock %.2f GHz\n",
         name, WAVES, E, ms, ops / (ms * 1e-3) / 1e12, ops / (ms * 1e-3) / 1e12 / 50.0, fa.numRegs, (int)fa.localSizeBytes,
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
  const int tiles = 3000;
  run<8, 32, 0, true>("32x32x32 regs", d, out, clk, tiles);
  run<8, 16, 0, true>("16x16x64 regs", d, out, clk, tiles);
  run<4, 32, 0, true>("32x32x32 regs", d, out, clk, tiles);
  run<4, 16, 0, true>("16x16x64 regs", d, out, clk, tiles);
  run<8, 32, 0, false>("32x32x32 lds", d, out, clk, tiles);
  run<8, 16, 0, false>("16x16x64 lds", d, out, clk, tiles);
  run<8, 32, 1, false>("32x32x32 lds", d, out, clk, tiles);
  run<8, 16, 1, false>("16x16x64 lds", d, out, clk, tiles);
  run<8, 32, 0, true>("32x32x32 regs (again)", d, out, clk, tiles);
  run<8, 16, 0, true>("16x16x64 regs (again)", d, out, clk, tiles);
  return 0;
}
