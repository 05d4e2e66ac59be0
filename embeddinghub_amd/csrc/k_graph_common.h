// Helpers shared by the graph-search kernels (k_graph.hip: the strict, hnswlib-order-identical walk; k_graphw.hip: the
// wide walk, several expansions per step).
#pragma once
#include "ehx_env.h"
#include "ehx_kernels.h"
#include "k_prep_query.h"

namespace ehx {
namespace {

constexpr uint32_t kNoNode = 0xFFFFFFFFu;

// -DEHX_GRAPH_PROFILE (ablation builds, scripts/gpu_graph_profile.sh): per-phase wall-clock ticks (100 MHz)
// of the level-0 loop, summed over all query waves into counters[4..11]: pick next node | adjacency +
// visited | row fetch + distances | rank fresh keys | decide next + request | insertion points | move R | tail.
// A/B switches of the level-0 loop (ablation builds only; the defaults are the shipped kernel)
#ifndef EHX_G_COOP
#define EHX_G_COOP 1        // rows read by 4-lane groups from the search copy (coalesced 64-byte pieces) instead of
#endif                      // one private row of X per lane
#ifndef EHX_G_NEXT_EARLY
#define EHX_G_NEXT_EARLY 1  // decide the next node before the merge and request its adjacency / visited words there
#endif

#ifndef EHX_G_WSYNC
#define EHX_G_WSYNC 1       // one wave per workgroup: LDS accesses of a wave execute in order, so a compiler-level
#endif                      // fence orders write -> read across lanes; no s_barrier, no drain of the LDS queue
#ifndef EHX_G_UNIFORM
#define EHX_G_UNIFORM 1     // wave-uniform values that come out of a shuffle or LDS are moved to scalar registers
#endif                      // (readfirstlane): the loop bookkeeping then runs on the scalar unit, with scalar branches

#if EHX_G_WSYNC
#define EHX_GSYNC() wave_lds_sync()
#else
#define EHX_GSYNC() __syncthreads()
#endif

#if EHX_G_UNIFORM
#define EHX_UNIFORM(x) wave_uniform((uint32_t)(x))
#else
#define EHX_UNIFORM(x) ((uint32_t)__shfl((int)(x), 0, 64))
#endif

#ifdef EHX_GRAPH_PROFILE
#define EHX_PROF_DECL unsigned long long prof_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, prof_t_ = wall_clock64()
#define EHX_PROF(i)                              \
  {                                              \
    const unsigned long long now_ = wall_clock64(); \
    prof_[i] += now_ - prof_t_;                  \
    prof_t_ = now_;                              \
  }
#else
#define EHX_PROF_DECL
#define EHX_PROF(i)
#endif

__device__ __forceinline__ uint64_t wave_sort64g(uint64_t key, int lane) {
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

// number of entries of the ascending array a[0..n) that are < key
// Prefetch-style load: a relaxed atomic load (wavefront scope: no cache-policy bits) is an ordered memory
// reference for the compiler, so it is issued where it is written — a plain load whose first use comes an
// LDS-heavy phase later is a candidate for the compiler's code sinking, which would expose the HBM round
// trip the early issue is meant to overlap.
__device__ __forceinline__ uint32_t load_here(const uint32_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}

// value of a 64-bit register in lane l (wave-uniform l)
__device__ __forceinline__ uint64_t readlane64(uint64_t v, int l) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l);
  return ((uint64_t)hi << 32) | lo;
}

__device__ __forceinline__ uint32_t lower_bound_lds(const uint64_t* a, uint32_t n, uint64_t key) {
  uint32_t lo = 0, hi = n;
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (a[mid] < key) lo = mid + 1;
    else hi = mid;
  }
  return lo;
}

}  // namespace
}  // namespace ehx
