// Microbenchmark: sustained v_mfma_f32_32x32x16_f16 rate with W waves per CU and A independent
// accumulators per wave, registers only.  Calibrates the "peak" the fp16 filter scan is compared to
// (the 2.5 PFLOP/s dense figure assumes the full boost clock; a long MFMA-dense kernel may not hold it).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int ACC>
__global__ __launch_bounds__(512) void mfma_loop(float* out, int iters, float a0, float b0) {
  f32x16 acc[ACC];
  for (int i = 0; i < ACC; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) {
    a[i] = (_Float16)(a0 + (threadIdx.x & 7) * 0.001f);
    b[i] = (_Float16)b0;
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int i = 0; i < ACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < ACC; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int ACC>
void run(int waves_per_cu, int iters) {
  int threads = waves_per_cu * 64, blocks = 256;
  float* out;
  hipMalloc(&out, blocks * threads * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  mfma_loop<ACC><<<blocks, threads>>>(out, 10, 0.001f, 0.002f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  mfma_loop<ACC><<<blocks, threads>>>(out, iters, 0.001f, 0.002f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  double flops = 2.0 * 32 * 32 * 16 * 16.0 * ACC * iters * (double)blocks * waves_per_cu;
  printf("waves/CU=%d acc/wave=%d: %.3f ms  %.1f TFLOP/s\n", waves_per_cu, ACC, ms, flops / ms / 1e9);
  hipFree(out);
}

int main() {
  run<4>(4, 8000);
  run<4>(8, 4000);
  run<4>(8, 40000);  // ~10x longer: does the clock hold?
  run<2>(8, 8000);
  run<1>(8, 16000);
  run<4>(16, 2000);
  return 0;
}
