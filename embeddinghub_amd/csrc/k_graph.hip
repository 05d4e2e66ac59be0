// Graph-mode search: HNSW-style greedy descent + level-0 best-first search over an HBM-resident
// graph, one wavefront per query.
//
// Replaces hnswlib::HierarchicalNSW::searchKnn / searchBaseLayerST (call site
// embeddinghub/embeddingstore/index.cc:41; algorithm restated in oracle/hnsw_oracle.hpp):
//   * upper levels maxlevel..1: greedy descent — scan the current node's list in stored order and
//     move to the FIRST strictly-closest neighbour, repeat until no improvement;
//   * level 0: best-first search bounded by ef.  hnswlib keeps two heaps (candidates, results) and a
//     lowerBound; here ONE sorted list R of at most ef (distance, id, expanded) keys lives in LDS:
//     the next node to expand is the closest unexpanded entry of R, the search ends when R has no
//     unexpanded entry.  This is equivalent to the two-heap formulation whenever distances are
//     distinct: every candidate is inserted into the result heap at the same moment, a candidate
//     evicted from the results has distance >= lowerBound and can never be expanded, and processing a
//     node's neighbours as one batch (R <- ef smallest of R u batch) keeps exactly the elements the
//     one-by-one insertion keeps.
//
// Layout re-designed for the GPU (SURVEY Appendix A.3 describes hnswlib's AoS element block):
//   * adj0[n][2M] u32, padded with 0xFFFFFFFF, stored order preserved: one 128-B line per expansion;
//   * upper levels: up_start[n] (index of the node's first upper list or ~0), up_lists[*][M];
//   * vectors: the space's row-major X plus a SEARCH COPY Xs (k_misc.hip: every 16-float block permuted
//     so that the four inputs of SSE partial sum j are contiguous; cosine rows normalised): a 4-lane
//     group reads a row in coalesced 64-byte pieces and lane j accumulates partial sum j — the canonical
//     (oracle-order) arithmetic, so on an imported graph the traversal, the returned ids and the
//     distances are bit-identical to the oracle;
//   * visited set: one bit per row per in-flight query in HBM (n/8 bytes per query — 1.25 MB at 10 M
//     rows, 1.3 GB for a 1024-query batch out of 288 GB), test-and-set with atomicOr.  The bitmap is all-zero
//     between launches: a query logs every row it marks (vislog, 48 ef + 256 entries) and clears exactly those
//     words when it is done — a few thousand stores instead of a 1.3-GB memset per batch (a query that outgrows
//     its log clears its whole bitmap instead);
//   * work counters as SURVEY §8d: n_dist = rows actually fetched, n_hops0 / n_hops_up = expansions.
#include "k_graph_common.h"

namespace ehx {

// LDS: q[ld] floats | R[ef_cap] u64 | S[64] u64 (sorted fresh keys) | batch[64] u64 | ids[64] u32 |
//      F[ef_cap] u8 (slots of R a fresh key lands on, during a merge)
// (the wide walk, k_graphw.hip, holds 32 ids per expansion of a step: ids[32 * width])
size_t graph_lds_bytes(uint32_t ld, uint32_t ef_cap, uint32_t width) {
  const size_t n_ids = width > 2 ? 32 * (size_t)width : 64;
  // (+ the wide walk's helper wave: hd[32] f32 + ctrl[2] u32)
  return (size_t)ld * 4 + (size_t)ef_cap * 8 + 64 * 8 * 2 + n_ids * 4 + (((size_t)ef_cap + 15) & ~(size_t)15) + 32 * 4 + 16 + 64;
}

// A/B builds: cap the registers so that this many waves share a SIMD (0 = the compiler's choice).  Measured in round 3:
// 3 and 4 make the compiler spill 79-535 registers per lane (it keeps the row walk's register rings alive across
// the LDS phases) — not shipped; at batch 1024 a SIMD holds one wave anyway.
#ifndef EHX_GRAPH_WAVES
#define EHX_GRAPH_WAVES 0
#endif
// (Round 4 built a HELPER wave per query — rows 16.. of every distance batch on a second SIMD, same arithmetic, bit-identical
// — and measured -1..2 % at batch 1024, +1..3 % at 2048 (profiles/r04_j_graph_*_helper{0,1}.jsonl): the row phase is bound
// by the memory system, not by the loads one wave keeps in flight.  Removed in round 6; the lever that pays at batch 1024 is
// fewer dependent steps per query: k_graphw.hip.  Round 6 re-measured the helper on THIS walk at short rows, where it does
// pay for the wide walk's 64-row passes: 6.25 M x 128, batch 1024, ef 50 / 200 / 800: 0.324 / 0.377 / 0.354 -> 0.294 / 0.344 /
// 0.331 of 8 TB/s (profiles/r06_l_graph_6250k128_help{0,1}.jsonl) — an expansion's <= 32 rows are ONE pass and one round
// trip either way, the split only adds two barriers.  Not in the library.)
template <int METRIC01, bool SCALE>
#if EHX_GRAPH_WAVES
__attribute__((amdgpu_waves_per_eu(EHX_GRAPH_WAVES, EHX_GRAPH_WAVES)))
#endif
__global__ __launch_bounds__(64) void graph_search_kernel(const GraphArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const uint32_t qi = blockIdx.x;
  float* qs = (float*)smem;
  uint64_t* R = (uint64_t*)(smem + (size_t)a.ld * 4);
  uint64_t* S = R + a.ef_cap;
  uint64_t* batch = S + 64;
  uint32_t* ids_l = (uint32_t*)(batch + 64);
  uint8_t* F = (uint8_t*)(ids_l + 64);
  uint32_t* vis = a.visited + (size_t)qi * a.vis_words;
  uint32_t* vlog = a.vislog + (size_t)qi * a.vislog_cap;
  uint32_t n_logged = 0;  // rows marked visited so far (wave-uniform)
  for (uint32_t i = lane; i < a.ef_cap; i += 64) F[i] = 0;

  if (a.q_raw) {
    // one query per call, one launch: the raw query comes from host-visible memory and is prepared here, by this wave,
    // into the device scratch row the loads below read (the same arithmetic as prep_queries_kernel: identical bytes)
    prep_query_row(a.q_raw, 1u, a.dims, a.ld, a.metric, const_cast<float*>(a.Q), 0u, lane);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  }
#if EHX_G_COOP
  for (uint32_t i = lane; i < a.ld; i += 64) qs[search_copy_pos(i)] = a.Q[(size_t)qi * a.ld + i];
#else
  for (uint32_t i = lane; i < a.ld; i += 64) qs[i] = a.Q[(size_t)qi * a.ld + i];
#endif
  EHX_GSYNC();

  unsigned long long n_dist = 0, n_hops0 = 0, n_hops_up = 0;

  // canonical distance of row ids_l[lane] for lane < count: every lane owns one neighbour row and
  // keeps the 4 SSE partial sums itself (16-byte loads; the query is an LDS broadcast read)
#if EHX_G_COOP
  // canonical distances of rows ids_l[0..count): 16 rows per pass, one 4-lane group per row reading the
  // search copy (canon_dist_group_t); lane p (< count) gets the distance of row p
  auto lane_dist = [&](uint32_t count) -> float {
    return wave_group_dists<METRIC01>(qs, a.Xs, a.ld, a.dims, ids_l, count, lane, a.xscale);
  };
#else
  auto lane_dist = [&](uint32_t count) -> float {
    if ((uint32_t)lane >= count) return __builtin_inff();
    const uint32_t id = ids_l[lane];
    const float xs = SCALE ? a.inv_norm[id] : 1.0f;
    return canon_dist_lane_t<METRIC01, SCALE>(qs, a.X + (size_t)id * a.ld, xs, a.dims);
  };
#endif

  // ---- entry point ----
  uint32_t cur = a.entry_point;
  if (lane == 0) ids_l[0] = cur;
  EHX_GSYNC();
  float curdist = __uint_as_float(EHX_UNIFORM(__float_as_uint(lane_dist(1))));
  n_dist += 1;

  // ---- upper levels: greedy descent ----
  for (int level = a.max_level; level >= 1; --level) {
    bool changed = true;
    while (changed) {
      changed = false;
      const uint32_t us = a.up_start[cur];
      const uint32_t* lst = a.up_lists + ((size_t)us + (uint32_t)(level - 1)) * a.M;
      uint32_t nb = kNoNode;
      if (lane < (int)a.M) nb = lst[lane];
      const uint64_t vmask = __ballot(nb != kNoNode);
      const uint32_t cnt = __builtin_popcountll(vmask);  // lists are packed from slot 0
      n_hops_up += 1;
      if (lane < (int)cnt) ids_l[lane] = nb;
      EHX_GSYNC();
      n_dist += cnt;
      uint32_t best_i = kNoNode;
      float best_d = curdist;
      {
        // first strictly-smaller minimum in stored order
        float m = lane_dist(cnt);
        uint32_t mi = (uint32_t)lane;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const float od = __shfl_xor(m, o, 64);
          const uint32_t oi = __shfl_xor(mi, o, 64);
          if (od < m || (od == m && oi < mi)) {
            m = od;
            mi = oi;
          }
        }
        m = __uint_as_float(EHX_UNIFORM(__float_as_uint(m)));  // (the butterfly leaves the minimum in every lane)
        mi = EHX_UNIFORM(mi);
        if (m < best_d) {
          best_d = m;
          best_i = mi;
        }
      }
      if (best_i != kNoNode) {
        curdist = best_d;
        cur = EHX_UNIFORM(ids_l[best_i]);
        changed = true;
      }
      EHX_GSYNC();
    }
  }

  // ---- level 0: best-first, ef bounded ----
  // Per expansion the dependent chain is: adjacency row -> visited words -> neighbour rows -> merge.
  //  * The adjacency row of the node most likely to be expanded next (c2, the second-closest
  //    unexpanded entry of R) is requested before this expansion's row fetches.  Right after the
  //    distances — BEFORE the merge — the next node is known for certain (c2, or the closest fresh
  //    neighbour if that is closer): c2's visited words, or the fresh node's adjacency row and then
  //    its visited words, are requested there and fly while R is merged.  On a confirmed c2 only the
  //    row fetch is left on the chain.  The traversal order is unchanged.
  //  * visited: the test is an agent-scope atomic LOAD, the set a fire-and-forget atomicOr (an
  //    adjacency list holds distinct ids — hnswlib invariant, checked on import — so the lanes of one
  //    expansion never race on the same BIT, and the same wave's later loads of a word observe its
  //    earlier atomics: per-location coherence).  The speculative words are loaded after this
  //    expansion's atomicOrs in program order and nothing else touches the bitmap before they are used.
  //  * merge: the fresh keys are ranked by counting (broadcast LDS reads, no shuffle network), their
  //    insertion points found by binary search, and R is updated IN PLACE from the top down, touching
  //    only [first insertion point, nR), one ballot + prefix popcount per 64 slots: nothing at all when
  //    no fresh key beats the current worst.
  const uint32_t ef = a.ef;
  uint32_t nR = 1;
  if (lane == 0) {
    R[0] = ((uint64_t)f32_to_ordered(curdist) << 32) | ((uint64_t)cur << 1);
    atomicOr(&vis[cur >> 5], 1u << (cur & 31));
    if (a.vislog_cap) vlog[0] = cur;
  }
  n_logged = 1;
  EHX_GSYNC();
  uint32_t scan_from = 0;  // every entry of R before this index is expanded
  uint32_t pf_node = kNoNode, pf_nb = kNoNode, pf_word = 0;
  unsigned long long n_pf_hit = 0;
  EHX_PROF_DECL;
  for (;;) {
    // closest unexpanded entry (and the one after it): 128 entries per trip (both LDS reads in flight
    // together), the two keys taken out of the registers with readlane — one LDS latency per trip
    uint32_t idx = kNoNode, idx2 = kNoNode;
    uint64_t kidx = 0, kidx2 = kKeyInf;
    for (uint32_t base = scan_from & ~63u; base < nR && idx2 == kNoNode; base += 128) {
      const uint32_t i0 = base + lane, i1 = i0 + 64;
      const uint64_t v0 = i0 < nR ? R[i0] : 1ull;  // beyond nR: "expanded"
      const uint64_t v1 = i1 < nR ? R[i1] : 1ull;
      uint64_t m0 = __ballot(!(v0 & 1ull)), m1 = __ballot(!(v1 & 1ull));
#pragma unroll
      for (int pick = 0; pick < 2; ++pick) {
        if (pick == 0 ? idx != kNoNode : (idx == kNoNode || idx2 != kNoNode)) continue;
        uint32_t at = kNoNode;
        uint64_t key = 0;
        if (m0) {
          const int l = __builtin_ctzll(m0);
          m0 &= m0 - 1;
          at = base + (uint32_t)l;
          key = readlane64(v0, l);
        } else if (m1) {
          const int l = __builtin_ctzll(m1);
          m1 &= m1 - 1;
          at = base + 64 + (uint32_t)l;
          key = readlane64(v1, l);
        }
        if (at == kNoNode) continue;
        if (pick == 0) {
          idx = at;
          kidx = key;
        } else {
          idx2 = at;
          kidx2 = key;
        }
      }
    }
    if (idx == kNoNode) break;
    EHX_PROF(0)
    const uint32_t c = (uint32_t)(kidx & 0xFFFFFFFFull) >> 1;
    const uint32_t c2 = idx2 != kNoNode ? (uint32_t)(kidx2 & 0xFFFFFFFFull) >> 1 : kNoNode;
    EHX_GSYNC();
    if (lane == 0) R[idx] |= 1ull;
    n_hops0 += 1;
    // neighbours (stored order) and their visited words
    uint32_t nb = kNoNode, word = 0;
    if (c == pf_node) {
      nb = pf_nb;
      word = pf_word;
    } else {  // first expansion only
      if (lane < (int)a.M0) nb = a.adj0[(size_t)c * a.M0 + lane];
      if (nb != kNoNode) word = __hip_atomic_load(&vis[nb >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const bool fresh = nb != kNoNode && !(word & (1u << (nb & 31)));
    if (fresh) (void)__hip_atomic_fetch_or(&vis[nb >> 5], 1u << (nb & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    pf_node = c2;
    pf_nb = kNoNode;
    if (c2 != kNoNode && lane < (int)a.M0) pf_nb = load_here(a.adj0 + (size_t)c2 * a.M0 + lane);
    const uint64_t fmask = __ballot(fresh);
    const uint32_t nfresh = __builtin_popcountll(fmask);
    if (fresh) {
      const uint32_t slot = (uint32_t)__builtin_popcountll(fmask & ((1ull << lane) - 1ull));
      ids_l[slot] = nb;
      if (n_logged + slot < a.vislog_cap) vlog[n_logged + slot] = nb;
    }
    n_logged += nfresh;
    EHX_GSYNC();
    n_dist += nfresh;
    EHX_PROF(1)
    // distances: lane p (< nfresh) owns fresh neighbour p
    uint64_t mykey = kKeyInf;
    if (nfresh) {
      const float d = lane_dist(nfresh);
      if ((uint32_t)lane < nfresh) mykey = ((uint64_t)f32_to_ordered(d) << 32) | ((uint64_t)ids_l[lane] << 1);
    }
    scan_from = idx2 != kNoNode ? idx2 : nR;  // entries before idx2 are all expanded now (positions only grow)
#ifdef EHX_GRAPH_PROFILE
    if (__any(mykey == 1ull)) prof_[7] += 1;  // keeps the distances live: the timer below waits for them
#endif
    EHX_PROF(2)
    // Does any fresh key enter R?  If so rank the fresh keys among themselves by counting (keys are
    // distinct: the id is part of the key); the key of rank 0 is the closest fresh neighbour.
    const bool do_merge = nfresh != 0 && (nR < ef || __any(mykey < R[ef - 1]));
    uint64_t minkey = kKeyInf;
    uint32_t rank = 0;
    if (do_merge) {
      batch[lane] = mykey;
      EHX_GSYNC();
      // sixteen keys per trip, all LDS reads issued before the first compare (a one-key-per-trip loop
      // pays the LDS latency nfresh times); batch[nfresh..64) = +inf never counts
      for (uint32_t j = 0; j < nfresh; j += 16) {
        uint64_t kb[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) kb[u] = batch[j + u];
#pragma unroll
        for (int u = 0; u < 16; ++u) rank += kb[u] < mykey ? 1u : 0u;
      }
      const uint64_t first = __ballot((uint32_t)lane < nfresh && rank == 0);
      minkey = readlane64(mykey, (int)__builtin_ctzll(first));
    }
    EHX_PROF(3)
    // The node expanded next is known NOW, before the merge: the closer of the closest fresh neighbour
    // and the second unexpanded entry c2 (a fresh key below R[idx2] is always inserted; one above it
    // leaves R[idx2] where it is).  Its adjacency row / visited words fly while R is merged.
    const uint64_t k2 = kidx2;  // key of the second unexpanded entry (+inf if there is none)
    bool pf_have_word;
#if EHX_G_NEXT_EARLY
    if (minkey < k2) {
      pf_node = (uint32_t)(minkey & 0xFFFFFFFFull) >> 1;
      pf_nb = kNoNode;
      if (lane < (int)a.M0) pf_nb = load_here(a.adj0 + (size_t)pf_node * a.M0 + lane);
      pf_have_word = false;
    } else
#endif
    {
      // c2 it is (EHX_G_NEXT_EARLY=0: presumably), and its adjacency row has landed (requested before the
      // row fetches): its visited words, loaded after this expansion's atomicOrs in program order
      pf_word = 0;
      if (pf_nb != kNoNode) pf_word = __hip_atomic_load(&vis[pf_nb >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      pf_have_word = true;
      if (!(minkey < k2)) n_pf_hit += 1;
    }
    EHX_PROF(4)
    if (do_merge) {
      if ((uint32_t)lane < nfresh) S[rank] = mykey;
      EHX_GSYNC();
      uint64_t skey = kKeyInf;
      uint32_t ps = kNoNode;
      if ((uint32_t)lane < nfresh) {
        skey = S[lane];  // the lane-th smallest fresh key
        ps = lower_bound_lds(R, nR, skey);
      }
      // insertion point of the smallest fresh key (lane 0).  readfirstlane, not a shuffle: the value is wave-uniform
      // and everything derived from it (nR, the scan positions, the loop bounds) then lives in scalar registers
      const uint32_t p0 = EHX_UNIFORM(ps);
      EHX_PROF(5)
      if (p0 < ef) {
        // Merge in place, driven by the DESTINATION: fresh key i lands at fpos = ps_i + i (distinct,
        // ascending); a destination slot no fresh key lands on receives the old entry whose index is the
        // slot minus the number of fresh keys landing below it.  F flags the landing slots, so per 64 slots
        // that number is one ballot + a lane-prefix popcount — no per-entry search.  Top down: a chunk
        // reads only slots at or below its own, which are still untouched.
        const uint32_t new_nR = nR + nfresh < ef ? nR + nfresh : ef;
        const uint32_t fpos = ps + (uint32_t)lane;
        const bool lands = (uint32_t)lane < nfresh && fpos < ef;
        if (lands) F[fpos] = 1;
        EHX_GSYNC();
        for (uint32_t dhi = new_nR; dhi > p0;) {
          const uint32_t dlo = dhi - p0 > 64 ? dhi - 64 : p0;
          const uint32_t dpos = dlo + (uint32_t)lane;
          const bool in = dpos < dhi;
          const bool taken = in && F[dpos] != 0;
          const uint64_t occ = __ballot(taken);
          const uint32_t below = (uint32_t)__builtin_popcountll(__ballot(lands && fpos < dlo));
          const uint32_t cnt = below + (uint32_t)__builtin_popcountll(occ & ((1ull << lane) - 1ull));
          const bool mv = in && !taken;
          uint64_t kj = 0;
          if (mv) kj = R[dpos - cnt];
          EHX_GSYNC();
          if (mv) R[dpos] = kj;
          EHX_GSYNC();
          dhi = dlo;
        }
        if (lands) {
          R[fpos] = skey;
          F[fpos] = 0;
        }
        EHX_GSYNC();
        nR = new_nR;
        if (p0 < scan_from) scan_from = p0;
      }
      EHX_PROF(6)
    }
    if (!pf_have_word) {
      // the adjacency row of the fresh node expanded next landed during the merge
      pf_word = 0;
      if (pf_nb != kNoNode) pf_word = __hip_atomic_load(&vis[pf_nb >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    EHX_PROF(7)
  }

  // ---- leave the visited bitmap all-zero: clear the words of the logged rows (or everything, if the log overflowed)
  // (the log was written by other lanes, through global memory: a real fence, once per query)
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  if (a.vislog_cap == 0) {
    // (A/B mode: the host clears the bitmaps with a memset before every launch)
  } else if (n_logged <= a.vislog_cap) {
    for (uint32_t i = lane; i < n_logged; i += 64) vis[vlog[i] >> 5] = 0u;
  } else {
    for (uint32_t i = lane; i < a.vis_words; i += 64) vis[i] = 0u;
  }
  // ---- results: the k closest of R (already sorted by (dist, id)) ----
  const uint32_t cnt = nR < a.k ? nR : a.k;
  for (uint32_t j = lane; j < a.k; j += 64) {
    const bool ok = j < cnt;
    a.out_ids[(size_t)qi * a.k + j] = ok ? (uint64_t)((uint32_t)(R[j] & 0xFFFFFFFFull) >> 1) : ~0ull;
    a.out_dist[(size_t)qi * a.k + j] = ok ? ordered_to_f32((uint32_t)(R[j] >> 32)) : __builtin_inff();
  }
  if (lane == 0) {
    a.out_count[qi] = cnt;
    atomicAdd(&a.counters[0], n_dist);
    atomicAdd(&a.counters[1], n_hops0);
    atomicAdd(&a.counters[2], n_hops_up);
    atomicAdd(&a.counters[3], n_pf_hit);
#ifdef EHX_GRAPH_PROFILE
    for (int i = 0; i < 8; ++i) atomicAdd(&a.counters[4 + i], prof_[i]);
#endif
  }
  if (a.done_flag) {   // one-launch form: the results above went to host-visible memory; tell the spinning host thread
    __threadfence_system();
    __builtin_amdgcn_s_waitcnt(0);
    if (lane == 0) __hip_atomic_store(a.done_flag, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

hipError_t launch_graph_search_wide(const GraphArgs& a, hipStream_t st);  // k_graphw.hip

hipError_t launch_graph_search(const GraphArgs& a, hipStream_t st) {
  // the wide walk serves lists of up to 32 ids (M <= 16: one half wave per expanded node)
  if (a.width > 1 && a.M0 <= 32) return launch_graph_search_wide(a, st);
  const size_t lds = graph_lds_bytes(a.ld, a.ef_cap, 1);
  static DynLdsAttr attr;
  const void* fns[3] = {(const void*)graph_search_kernel<0, false>, (const void*)graph_search_kernel<1, true>,
                        (const void*)graph_search_kernel<1, false>};
  if (hipError_t e = attr.ensure(fns, 3, lds); e != hipSuccess) return e;
  if (a.metric == 0) hipLaunchKernelGGL((graph_search_kernel<0, false>), dim3(a.nq), dim3(64), lds, st, a);
  else if (a.metric == 2) hipLaunchKernelGGL((graph_search_kernel<1, true>), dim3(a.nq), dim3(64), lds, st, a);
  else hipLaunchKernelGGL((graph_search_kernel<1, false>), dim3(a.nq), dim3(64), lds, st, a);
  return hipGetLastError();
}

}  // namespace ehx
