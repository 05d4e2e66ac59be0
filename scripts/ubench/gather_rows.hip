// Microbenchmark: what HBM delivers for the graph path's access pattern — random whole-row gathers.
// Each wave repeatedly picks a pseudo-random row of `row_bytes` contiguous bytes out of a `pool_mb` MiB
// buffer (far beyond the 256 MB Infinity Cache for the large pools) and reads it (a) cooperatively
// (64 lanes x 16 B = 1 KiB per instruction: the best-case coalescing) or (b) lane-serially (every lane
// walks its OWN random row with 16-byte loads, `depth` loads in flight: the graph kernel's pattern).
// Reports GB/s vs waves per CU.  The streaming (sequential) rate of the same kernel shape is printed
// for reference.  Build: hipcc --offload-arch=gfx950 -O3 gather_rows.hip -o gather_rows.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

__device__ __forceinline__ uint64_t mix(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
  return x;
}

// cooperative: one row per wave per trip
__global__ __launch_bounds__(64) void gather_coop(const float4* __restrict__ pool, uint64_t n_rows, uint32_t row_f4,
                                                  int trips, int sequential, float* out) {
  const int lane = threadIdx.x;
  float4 acc = {0, 0, 0, 0};
  uint64_t seed = (uint64_t)blockIdx.x * 0x9E3779B97F4A7C15ull + 12345;
  for (int t = 0; t < trips; ++t) {
    const uint64_t r = sequential ? ((uint64_t)blockIdx.x * trips + t) % n_rows : mix(seed + t) % n_rows;
    const float4* row = pool + r * row_f4;
    for (uint32_t i = lane; i < row_f4; i += 64) {
      const float4 v = row[i];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  }
  out[blockIdx.x * 64 + lane] = acc.x + acc.y + acc.z + acc.w;
}

// lane-serial: every active lane walks its own random row, DEPTH 16-byte loads in flight
template <int DEPTH>
__global__ __launch_bounds__(64) void gather_lane(const float4* __restrict__ pool, uint64_t n_rows, uint32_t row_f4,
                                                  int trips, int active, float* out) {
  const int lane = threadIdx.x;
  float4 acc = {0, 0, 0, 0};
  uint64_t seed = ((uint64_t)blockIdx.x * 64 + lane) * 0x9E3779B97F4A7C15ull + 777;
  if (lane < active) {
    for (int t = 0; t < trips; ++t) {
      const uint64_t r = mix(seed + t) % n_rows;
      const float4* row = pool + r * row_f4;
      for (uint32_t i = 0; i < row_f4; i += DEPTH) {
        float4 v[DEPTH];
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) v[u] = row[i + u];
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
      }
    }
  }
  out[blockIdx.x * 64 + lane] = acc.x + acc.y + acc.z + acc.w;
}

int main(int argc, char** argv) {
  const size_t pool_mb = argc > 1 ? atol(argv[1]) : 8192;
  const uint32_t row_bytes = argc > 2 ? atoi(argv[2]) : 3072;
  const uint32_t row_f4 = row_bytes / 16;
  const uint64_t n_rows = pool_mb * 1048576ull / row_bytes;
  float4* pool;
  float* out;
  if (hipMalloc(&pool, n_rows * (size_t)row_bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(pool, 0, n_rows * (size_t)row_bytes);
  hipMalloc(&out, 256 * 64 * 64 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  printf("pool %zu MiB, rows of %u B (%llu rows)\n", pool_mb, row_bytes, (unsigned long long)n_rows);
  auto time = [&](auto launch, double bytes, const char* name, int wpc) {
    launch();  // warm-up
    hipDeviceSynchronize();
    hipEventRecord(e0);
    launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s waves/CU %2d : %8.1f GB/s  (%.3f ms)\n", name, wpc, bytes / ms / 1e6, ms);
  };
  for (int wpc : {4, 8, 16, 32}) {
    const int blocks = 256 * wpc;
    const int trips = 2048 / wpc * 4;
    time([&] { gather_coop<<<blocks, 64>>>(pool, n_rows, row_f4, trips, 1, out); }, (double)blocks * trips * row_bytes,
         "cooperative, sequential rows", wpc);
    time([&] { gather_coop<<<blocks, 64>>>(pool, n_rows, row_f4, trips, 0, out); }, (double)blocks * trips * row_bytes,
         "cooperative, random rows", wpc);
  }
  for (int wpc : {4, 8}) {
    const int blocks = 256 * wpc;
    const int trips = 64 / wpc * 4;
    const int active = 27;  // mean fresh neighbours per expansion in the graph bench
    time([&] { gather_lane<16><<<blocks, 64>>>(pool, n_rows, row_f4, trips, active, out); },
         (double)blocks * trips * active * row_bytes, "lane-serial x27, 16 loads in flight", wpc);
    if (row_f4 % 48) continue;
    time([&] { gather_lane<48><<<blocks, 64>>>(pool, n_rows, row_f4, trips, active, out); },
         (double)blocks * trips * active * row_bytes, "lane-serial x27, 48 loads in flight", wpc);
    time([&] { gather_lane<48><<<blocks, 64>>>(pool, n_rows, row_f4, trips, 64, out); },
         (double)blocks * trips * 64 * row_bytes, "lane-serial x64, 48 loads in flight", wpc);
  }
  return 0;
}
