// Pieces shared by the 8-wave scan kernels (k_flat8.hip: fp32 MFMA scan; k_flat16.hip: fp16 MFMA
// filter scan): global->LDS DMA, barriers, the wave-local candidate lists.
#pragma once
#include "ehx_kernels.h"

namespace ehx {
namespace {

constexpr uint32_t kLists8 = 512;  // candidate lists per workgroup: one per (wave, query of the wave)

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ void glds16_8(const void* gsrc, void* lds_dst_uniform) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_dst_uniform, 16, 0, 0);
}

__device__ __forceinline__ uint64_t wave_sort64_8(uint64_t key, int lane) {
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

// hot-path barrier: LDS traffic only, the DMA queue keeps flowing
__device__ __forceinline__ void hot_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
// cold-path barrier: also publishes this wave's global stores (candidate slots) to the workgroup
__device__ __forceinline__ void cold_barrier() {
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// append (score,id) to query q's candidate slots.  Returns the slot position (>= kCandSlots: the list
// was full and nothing was stored) or -1 when the key does not beat the query's threshold.
__device__ __attribute__((noinline)) int scan8_push(float sc, uint32_t grow, int list, uint32_t n, uint64_t* cand,
                                                    int* cnt, const uint64_t* thr_key) {
  const uint64_t key = ((uint64_t)f32_to_ordered(sc) << 32) | grow;
  if (grow < n && key < thr_key[list]) {
    const int pos = atomicAdd(&cnt[list], 1);
    if (pos < (int)kCandSlots) cand[list * kCandSlots + pos] = key;
    return pos;
  }
  return -1;
}

// Wave-local compaction of this wave's own 64 candidate lists (list = w*64 + local query): sort the
// slots, keep the best kprime, tighten the list's threshold and share it with the other chunks'
// workgroups.  Only the owning wave ever touches a list, so no workgroup barrier is involved; the
// wave's own slot stores are made visible to itself with vmcnt(0) (this also drains its DMA pieces —
// rare, and the sibling wave keeps the SIMD's matrix pipe busy meanwhile).
__device__ __attribute__((noinline)) void scan8_compact(int w, int wc, int lane, int kprime, bool force,
                                                        uint64_t* cand, int* cnt, uint64_t* thr_key, float* thr_f,
                                                        unsigned long long* gthr_tile) {
  const int trigger = kprime + ((int)kCandSlots - kprime) / 2;
  const int c = cnt[w * 64 + lane];
  uint64_t mask = __ballot(force ? c > 0 : c >= trigger);
  if (!mask) return;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  while (mask) {
    const int ql = __builtin_ctzll(mask);
    mask &= mask - 1;
    const int list = w * 64 + ql;
    const int cq = cnt[list];
    const int nv = cq < (int)kCandSlots ? cq : (int)kCandSlots;
    uint64_t key = lane < nv ? cand[list * kCandSlots + lane] : kKeyInf;
    key = wave_sort64_8(key, lane);
    if (lane < kprime) cand[list * kCandSlots + lane] = key;
    const uint64_t kth = __shfl(key, kprime - 1, 64);
    if (lane == 0) {
      cnt[list] = nv < kprime ? nv : kprime;
      if (nv >= kprime) {
        // this list's k'-th best bounds the GLOBAL k'-th best of the query from above: share it
        // (monotone atomicMin; any stale value is safe to filter with)
        const int q = wc * 64 + ql;
        const unsigned long long old = atomicMin(&gthr_tile[q], (unsigned long long)kth);
        const uint64_t best = old < kth ? old : kth;
        thr_key[list] = best;
        thr_f[list] = ordered_to_f32((uint32_t)(best >> 32));
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
}

}  // namespace
}  // namespace ehx
