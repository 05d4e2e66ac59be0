// Microbenchmark 2: the scan kernel's MFMA stream shape without memory: 8 waves per CU, 2x2 blocks,
// groups of 16 MFMAs whose A/B operands come from 4-wide "fragments" that are refreshed per group.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA(A, B, C) __builtin_amdgcn_mfma_f32_32x32x2f32((A), (B), (C), 0, 0, 0)

template <int VARIANT>
__global__ __launch_bounds__(512, 2) void stream(float* out, int stages, float seed) {
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2; ++j)
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  f32x4 fa[2], fb[2];
  for (int i = 0; i < 2; ++i) {
    fa[i] = (f32x4){seed, seed + 1, seed + 2, seed + 3};
    fb[i] = (f32x4){seed, seed - 1, seed - 2, seed - 3};
  }
  for (int s = 0; s < stages; ++s) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if (VARIANT >= 1) {  // fragments change per group (opaque to the compiler)
        asm volatile("" : "+v"(fa[0]), "+v"(fa[1]), "+v"(fb[0]), "+v"(fb[1]));
      }
      if (VARIANT == 5) {
        f32x16 t00 = MFMA(fa[0][0], fb[0][0], acc[0][0]);
        f32x16 t10 = MFMA(fa[1][0], fb[0][0], acc[1][0]);
        f32x16 t01 = MFMA(fa[0][0], fb[1][0], acc[0][1]);
        f32x16 t11 = MFMA(fa[1][0], fb[1][0], acc[1][1]);
        asm volatile("" : "+v"(t00), "+v"(t10), "+v"(t01), "+v"(t11));
#pragma unroll
        for (int tt = 1; tt < 4; ++tt) {
          t00 = MFMA(fa[0][tt], fb[0][tt], t00);
          t10 = MFMA(fa[1][tt], fb[0][tt], t10);
          t01 = MFMA(fa[0][tt], fb[1][tt], t01);
          t11 = MFMA(fa[1][tt], fb[1][tt], t11);
        }
        acc[0][0] = t00; acc[1][0] = t10; acc[0][1] = t01; acc[1][1] = t11;
        continue;
      }
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) {
        acc[0][0] = MFMA(fa[0][tt], fb[0][tt], acc[0][0]);
        acc[1][0] = MFMA(fa[1][tt], fb[0][tt], acc[1][0]);
        acc[0][1] = MFMA(fa[0][tt], fb[1][tt], acc[0][1]);
        acc[1][1] = MFMA(fa[1][tt], fb[1][tt], acc[1][1]);
        if (VARIANT >= 2 && (tt & 1)) __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (VARIANT >= 3) {  // one barrier per stage
      __builtin_amdgcn_s_barrier();
    }
    if (VARIANT >= 4) {  // scalar bookkeeping + data-dependent uniform branches like the real loop
      if (__builtin_amdgcn_readfirstlane((int)(s * 2654435761u) >> 28) == 3) {
        asm volatile("" : "+v"(fa[0]));
      }
    }
  }
  float sum = 0;
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2; ++j)
      for (int r = 0; r < 16; ++r) sum += acc[i][j][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
}

template <int V>
void run(const char* name) {
  const int blocks = 256, threads = 512, stages = 4000;
  float* out;
  (void)hipMalloc(&out, blocks * threads * 4);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  stream<V><<<blocks, threads>>>(out, 10, 1.f);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  stream<V><<<blocks, threads>>>(out, stages, 1.f);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  double flops = 2.0 * 32 * 32 * 2 * 64.0 * stages * (double)blocks * 8;
  printf("%-44s %.3f ms  %.1f TFLOP/s\n", name, ms, flops / ms / 1e9);
  (void)hipFree(out);
}

int main() {
  run<0>("same fragments");
  run<1>("fragments refreshed per group");
  run<2>("+ sched_barrier every 8 MFMAs");
  run<3>("+ s_barrier per stage (64 MFMAs/wave)");
  run<4>("+ uniform branch per stage");
  run<5>("dst != srcC (accumulators ping-pong)");
  return 0;
}
