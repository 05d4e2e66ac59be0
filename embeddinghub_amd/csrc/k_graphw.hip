// Graph-mode search, WIDE walk: the level-0 best-first search of k_graph.hip with P = 2 or 4 expansions per step
// (ehx_params.search_width; opt-in — the strict walk stays the default and stays hnswlib's order).
//
// Why: at the batch the metric is quoted on (1024 queries = one wave per SIMD) the strict walk is a chain of dependent
// steps — adjacency row -> visited words -> neighbour rows -> merge of the result list — and every issued instruction and
// every memory round trip of a step is on the critical path of its query (k_graph.hip: ~700 instructions and 1.5-2 round
// trips per expansion; 0.31-0.37 of the HBM peak on 128-dim rows with the machine half empty).  hnswlib expands ONE
// candidate per step because a CPU thread has nothing to gain from more; a wave has: here a step takes the P closest
// unexpanded entries of the result list R together —
//   * their P adjacency rows are one load instruction (half a wave per list of <= 32 ids),
//   * one atomic test-and-set per neighbour on the visited bitmap (fetch_or: two lists of a step may name the same row —
//     exactly one lane sees the bit clear),
//   * all <= 32 P fresh rows go through the distance passes back to back, and R is merged once per 64 fresh keys,
//   * the nodes of the NEXT step are predicted before R is moved (the P smallest of: the next P unexpanded entries of R
//     and the fresh keys that enter it) and their adjacency rows fly during the merge; the pick after the merge is
//     authoritative and only looks the prediction up —
// so the pick / rank / insert / move phases and the round trips are paid once per P expansions.  The ef bound, the
// termination rule (no unexpanded entry left in R) and the result (the k closest of R, canonical distances, (distance, id)
// order) are the strict walk's; what changes is the ORDER of expansions: a node is expanded although a closer one might
// have been found by expanding its sibling first, so a query fetches a few per cent more rows (n_dist) and its result may
// differ from hnswlib's in the tail.  The gate (tests/test_graph_wide.py): recall@10 within 0.005 of the
// strict walk / the oracle's HNSW at equal ef, rows fetched within +15 %.
//
// Replaces hnswlib::HierarchicalNSW::searchKnn (call site embeddinghub/embeddingstore/index.cc:41) as a throughput mode;
// layout, visited bitmaps, visit log and counters as k_graph.hip.
#include "k_graph_common.h"

namespace ehx {

// HELP (short rows, batches that leave SIMDs empty): a second wave per query takes rows 32.. of every 64-row distance pass.
// A wave's row phase runs at ~10 GB/s whatever it keeps in flight (e.6 in DESIGN.md: a pass of 64 random 512-byte rows takes
// 2.8 us, one of 32 rows 1.4 us — an address translation per row), and at batch 1024 every SIMD holds one wave: two waves
// fetch a step's rows side by side.  Same 4-lane-group arithmetic on the same rows: distances, traversal and counters are
// bit-identical to the one-wave form.  Protocol: wave 0 publishes (first row slot, count), barrier, both compute, barrier,
// wave 0 reads the helper's distances from LDS; everything else (R, visited set, merges) is wave 0's alone.  (Round 4 tried
// the same on the strict walk at 3-KB rows, where the memory system — not the wave — bounds the row phase: no gain.)
template <int METRIC01, int P, bool HELP>
__global__ __launch_bounds__(HELP ? 128 : 64) void graph_search_wide_kernel(const GraphArgs a) {
  constexpr int NREG = P / 2;  // adjacency registers per lane: slot 64 r + lane = entry (lane & 31) of node 2 r + (lane >> 5)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint32_t qi = blockIdx.x;
  float* qs = (float*)smem;
  uint64_t* R = (uint64_t*)(smem + (size_t)a.ld * 4);
  uint64_t* S = R + a.ef_cap;
  uint64_t* batch = S + 64;
  uint32_t* ids_l = (uint32_t*)(batch + 64);
  uint8_t* F = (uint8_t*)(ids_l + 32 * P);
  float* hd = (float*)(F + (((size_t)a.ef_cap + 15) & ~(size_t)15));   // HELP: the helper wave's distances [32]
  volatile uint32_t* ctrl = (volatile uint32_t*)(hd + 32);               // HELP: (count, first slot) of the pass; ~0: done
  uint32_t* pki = (uint32_t*)batch;  // indices of a step's picks (batch[] is free outside the rank phase)
  uint32_t* vis = a.visited + (size_t)qi * a.vis_words;
  uint32_t* vlog = a.vislog + (size_t)qi * a.vislog_cap;
  uint32_t n_logged = 0;
  if (wv == 0)
    for (uint32_t i = lane; i < a.ef_cap; i += 64) F[i] = 0;

  if (wv == 0 && a.q_raw) {  // one query per call in one launch (k_graph.hip)
    prep_query_row(a.q_raw, 1u, a.dims, a.ld, a.metric, const_cast<float*>(a.Q), 0u, lane);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  }
  if (HELP) __syncthreads();   // (the prepared query row is wave 0's work)
  for (uint32_t i = threadIdx.x; i < a.ld; i += (HELP ? 128 : 64)) qs[search_copy_pos(i)] = a.Q[(size_t)qi * a.ld + i];
  if (HELP) __syncthreads();
  else EHX_GSYNC();
  if (HELP && wv == 1) {   // the helper wave: rows 32.. of every pass wave 0 publishes
    for (;;) {
      __syncthreads();  // A: (count, first slot) are published
      const uint32_t cnt = ctrl[0];
      if (cnt == 0xFFFFFFFFu) break;
      const uint32_t f0 = ctrl[1];
      if (cnt > 32) {
        const float d = wave_group_dists<METRIC01>(qs, a.Xs, a.ld, a.dims, ids_l + f0 + 32, cnt - 32, lane, a.xscale);
        if ((uint32_t)lane < cnt - 32) hd[lane] = d;
      }
      __syncthreads();  // B: the distances are published
    }
    return;
  }

  unsigned long long n_dist = 0, n_hops0 = 0, n_hops_up = 0, n_steps = 0, n_pf_hit = 0;
  const uint64_t lt_mask = (1ull << lane) - 1ull;

  // ---- entry point and upper levels: the strict walk's greedy descent ----
  uint32_t cur = a.entry_point;
  if (lane == 0) ids_l[0] = cur;
  EHX_GSYNC();
  float curdist = __uint_as_float(
      EHX_UNIFORM(__float_as_uint(wave_group_dists<METRIC01>(qs, a.Xs, a.ld, a.dims, ids_l, 1, lane, a.xscale))));
  n_dist += 1;
  for (int level = a.max_level; level >= 1; --level) {
    bool changed = true;
    while (changed) {
      changed = false;
      const uint32_t us = a.up_start[cur];
      const uint32_t* lst = a.up_lists + ((size_t)us + (uint32_t)(level - 1)) * a.M;
      uint32_t nb = kNoNode;
      if (lane < (int)a.M) nb = lst[lane];
      const uint32_t cnt = __builtin_popcountll(__ballot(nb != kNoNode));
      n_hops_up += 1;
      if (lane < (int)cnt) ids_l[lane] = nb;
      EHX_GSYNC();
      n_dist += cnt;
      float m = wave_group_dists<METRIC01>(qs, a.Xs, a.ld, a.dims, ids_l, cnt, lane, a.xscale);
      uint32_t mi = (uint32_t)lane;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {  // first strictly-smaller minimum in stored order
        const float od = __shfl_xor(m, o, 64);
        const uint32_t oi = __shfl_xor(mi, o, 64);
        if (od < m || (od == m && oi < mi)) {
          m = od;
          mi = oi;
        }
      }
      m = __uint_as_float(EHX_UNIFORM(__float_as_uint(m)));
      mi = EHX_UNIFORM(mi);
      if (m < curdist) {
        curdist = m;
        cur = EHX_UNIFORM(ids_l[mi]);
        changed = true;
      }
      EHX_GSYNC();
    }
  }

  // ---- level 0: best-first, ef bounded, P expansions per step ----
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
  uint32_t pf_node[P];     // nodes whose adjacency rows were requested at the end of the last step (wave-uniform)
  uint32_t pf_nb[NREG];    // ... and the rows, slot layout as above
#pragma unroll
  for (int j = 0; j < P; ++j) pf_node[j] = kNoNode;
#pragma unroll
  for (int r = 0; r < NREG; ++r) pf_nb[r] = kNoNode;
  const uint32_t e_l = (uint32_t)lane & 31u;
  const bool hi_l = lane >= 32;

  // -DEHX_GRAPH_PROFILE builds: 100-MHz ticks per phase summed over all waves into counters[5..11]:
  // pick | adjacency + visited + compaction | row fetch + distances | rank + prediction | insertion points | move R | rest
  EHX_PROF_DECL;
  for (;;) {
    EHX_PROF(6)
    // (Measured and dropped: touching the visited words of the predicted nodes' neighbours here — an atomic OR of 0, nothing to
    // wait for — so that the test-and-set after the pick finds its lines in the L2.  6.25 M x 128, same box, ef 50 / 200 / 800:
    // +7 % time at either width (profiles/r06_e_graph_6250k128_{warm,nowarm}.jsonl): the extra atomic per neighbour costs
    // more than the shorter round trip saves.)
    // ---- pick: the first P unexpanded entries of R (marked expanded on the spot) and the P after them (the old entries
    // the next step can pick from); 128 entries per trip, keys and positions through S / pki ----
    uint32_t found = 0;
    for (uint32_t base = scan_from & ~63u; base < nR && found < 2 * P; base += 128) {
      const uint32_t i0 = base + lane, i1 = i0 + 64;
      const uint64_t v0 = i0 < nR ? R[i0] : 1ull;  // beyond nR: "expanded"
      const uint64_t v1 = i1 < nR ? R[i1] : 1ull;
      const bool u0 = !(v0 & 1ull), u1 = !(v1 & 1ull);
      const uint64_t m0 = __ballot(u0), m1 = __ballot(u1);
      const uint32_t c0 = (uint32_t)__builtin_popcountll(m0);
      const uint32_t pre0 = found + (uint32_t)__builtin_popcountll(m0 & lt_mask);
      const uint32_t pre1 = found + c0 + (uint32_t)__builtin_popcountll(m1 & lt_mask);
      if (u0 && pre0 < 2 * P) {
        S[pre0] = v0;
        pki[pre0] = i0;
        if (pre0 < P) R[i0] = v0 | 1ull;
      }
      if (u1 && pre1 < 2 * P) {
        S[pre1] = v1;
        pki[pre1] = i1;
        if (pre1 < P) R[i1] = v1 | 1ull;
      }
      found += c0 + (uint32_t)__builtin_popcountll(m1);
    }
    if (found == 0) break;
    EHX_GSYNC();
    const uint32_t npick = found < P ? found : P;
    const uint32_t nu = (found < 2 * P ? found : 2 * P) - npick;
    uint64_t pkv = kKeyInf;
    uint32_t pkiv = 0;
    if ((uint32_t)lane < npick + nu) {
      pkv = S[lane];
      pkiv = pki[lane];
    }
    uint32_t pnode[P];   // the step's nodes (wave-uniform)
    uint64_t ckey[P];    // candidates of the next step: the old unexpanded entries after the picks, ascending (+inf: none)
#pragma unroll
    for (int j = 0; j < P; ++j) {
      const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)pkv, j);
      pnode[j] = (uint32_t)j < npick ? lo >> 1 : kNoNode;
      ckey[j] = readlane64(pkv, P + j);  // (lanes >= npick + nu hold +inf)
    }
    scan_from = nu ? (uint32_t)__builtin_amdgcn_readlane((int)pkiv, P) : nR;
    EHX_GSYNC();
    n_hops0 += npick;
    n_steps += 1;
    EHX_PROF(0)

    // ---- adjacency rows: from the registers requested a step ago where the prediction named the node, else loaded ----
    int mj[P];
#pragma unroll
    for (int j = 0; j < P; ++j) {
      mj[j] = -1;
#pragma unroll
      for (int jj = 0; jj < P; ++jj)
        if (pnode[j] != kNoNode && pnode[j] == pf_node[jj]) mj[j] = jj;
      if (mj[j] >= 0) n_pf_hit += 1;
    }
    uint32_t nb[NREG];
#pragma unroll
    for (int r = 0; r < NREG; ++r) {
      const uint32_t node = hi_l ? pnode[2 * r + 1] : pnode[2 * r];
      const int m = hi_l ? mj[2 * r + 1] : mj[2 * r];
      uint32_t got = kNoNode;
#pragma unroll
      for (int rr = 0; rr < NREG; ++rr) {
        const uint32_t t = (uint32_t)__shfl((int)pf_nb[rr], (int)((((uint32_t)m & 1u) << 5) + e_l), 64);
        if ((m >> 1) == rr) got = t;
      }
      uint32_t v = kNoNode;
      if (node != kNoNode && e_l < a.M0) {
        if (m >= 0) v = got;
        else v = a.adj0[(size_t)node * a.M0 + e_l];
      }
      nb[r] = v;
    }
    // ---- visited: one atomic test-and-set per neighbour ----
    uint32_t old[NREG];
#pragma unroll
    for (int r = 0; r < NREG; ++r) {
      old[r] = 0xFFFFFFFFu;
      if (nb[r] != kNoNode)
        old[r] = __hip_atomic_fetch_or(&vis[nb[r] >> 5], 1u << (nb[r] & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    uint32_t nfresh = 0;
#pragma unroll
    for (int r = 0; r < NREG; ++r) {
      const bool fresh = nb[r] != kNoNode && !(old[r] & (1u << (nb[r] & 31)));
      const uint64_t fm = __ballot(fresh);
      if (fresh) {
        const uint32_t slot = nfresh + (uint32_t)__builtin_popcountll(fm & lt_mask);
        ids_l[slot] = nb[r];
        if (n_logged + slot < a.vislog_cap) vlog[n_logged + slot] = nb[r];
      }
      nfresh += (uint32_t)__builtin_popcountll(fm);
    }
    n_logged += nfresh;
    n_dist += nfresh;
    EHX_GSYNC();
    EHX_PROF(1)

    // the adjacency rows of the nodes the next step is expected to pick (ckey), requested once the last fresh key is ranked
    auto request_next = [&]() {
#pragma unroll
      for (int j = 0; j < P; ++j) pf_node[j] = ckey[j] == kKeyInf ? kNoNode : (uint32_t)(ckey[j] & 0xFFFFFFFFull) >> 1;
#pragma unroll
      for (int r = 0; r < NREG; ++r) {
        const uint32_t node = hi_l ? pf_node[2 * r + 1] : pf_node[2 * r];
        pf_nb[r] = kNoNode;
        if (node != kNoNode && e_l < a.M0) pf_nb[r] = load_here(a.adj0 + (size_t)node * a.M0 + e_l);
      }
    };
    if (nfresh == 0) request_next();

    // ---- distances of every fresh row (64 per pass), then the merge ----
    constexpr int NCH = P / 2;   // passes of 64 rows a step can need
    uint64_t key[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      key[c] = kKeyInf;
      if ((uint32_t)c * 64u < nfresh) {
        const uint32_t f0 = (uint32_t)c * 64u;
        const uint32_t cnt = nfresh - f0 < 64 ? nfresh - f0 : 64;
        float d;
        if (HELP) {
          if (lane == 0) {
            ctrl[0] = cnt;
            ctrl[1] = f0;
          }
          __syncthreads();  // A
          d = wave_group_dists<METRIC01>(qs, a.Xs, a.ld, a.dims, ids_l + f0, cnt < 32 ? cnt : 32, lane, a.xscale);
          __syncthreads();  // B
          if (lane >= 32 && (uint32_t)lane < cnt) d = hd[lane - 32];
        } else {
          d = wave_group_dists<METRIC01, true>(qs, a.Xs, a.ld, a.dims, ids_l + f0, cnt, lane, a.xscale);
        }
        if ((uint32_t)lane < cnt) key[c] = ((uint64_t)f32_to_ordered(d) << 32) | ((uint64_t)ids_l[f0 + lane] << 1);
      }
    }
    EHX_PROF(2)
    // one merge of up to 64 keys (lane p holds key p, +inf: none) into R; `last`: the step's last merge — the next step's
    // adjacency rows are requested inside it, once its keys are ranked
    auto merge_keys = [&](const uint64_t mykey, const bool last) {
      const uint64_t bound = nR < ef ? kKeyInf : R[ef - 1];
      const bool can = mykey < bound;  // (a key at or above the worst entry of a full list never enters)
      const bool do_merge = __any(can);
      if (!do_merge) {
        if (last) request_next();
        return;
      }
      // Only the keys that can enter go on: compacted into batch[0..ncan) (in the steady state of a search — R full —
      // most fresh keys are worse than R's worst entry, and the ranking below costs a trip per 16 keys).
      const uint64_t cmask = __ballot(can);
      const uint32_t ncan = (uint32_t)__builtin_popcountll(cmask);
      if ((uint32_t)lane >= ncan) batch[lane] = kKeyInf;
      if (can) batch[__builtin_popcountll(cmask & lt_mask)] = mykey;
      EHX_GSYNC();
      // rank of those keys among themselves by counting (k_graph.hip); a lane that holds none ranks nothing
      uint32_t rank = 0;
      for (uint32_t j = 0; j < ncan; j += 16) {
        uint64_t kb[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) kb[u] = batch[j + u];
#pragma unroll
        for (int u = 0; u < 16; ++u) rank += kb[u] < mykey ? 1u : 0u;
      }
      // prediction: ckey <- the P smallest of ckey u {fresh keys that can enter}.  A fresh key of rank r sits at
      // r + #(ckey below it) in the union, ckey[j] at j + #(fresh keys below it); keys are distinct (the id is in them).
      {
        uint32_t ur = rank;
        uint32_t cpos[P];
#pragma unroll
        for (int j = 0; j < P; ++j) {
          ur += ckey[j] < mykey ? 1u : 0u;
          cpos[j] = (uint32_t)j + (uint32_t)__builtin_popcountll(__ballot(can && mykey < ckey[j]));
        }
        uint64_t nkey[P];
#pragma unroll
        for (int t = 0; t < P; ++t) {
          nkey[t] = kKeyInf;
          const uint64_t bt = __ballot(can && ur == (uint32_t)t);
          if (bt) nkey[t] = readlane64(mykey, (int)__builtin_ctzll(bt));
#pragma unroll
          for (int j = 0; j < P; ++j)
            if (ckey[j] != kKeyInf && cpos[j] == (uint32_t)t) nkey[t] = ckey[j];
        }
#pragma unroll
        for (int t = 0; t < P; ++t) ckey[t] = nkey[t];
      }
      if (last) request_next();
      EHX_PROF(3)
      // merge into R in place, top down (k_graph.hip): the ncan keys, sorted through S
      if (can) S[rank] = mykey;
      EHX_GSYNC();
      uint64_t skey = kKeyInf;
      uint32_t ps = kNoNode;
      if ((uint32_t)lane < ncan) {
        skey = S[lane];
        ps = lower_bound_lds(R, nR, skey);
      }
      const uint32_t p0 = EHX_UNIFORM(ps);
      EHX_PROF(4)
      if (p0 < ef) {
        const uint32_t new_nR = nR + ncan < ef ? nR + ncan : ef;
        const uint32_t fpos = ps + (uint32_t)lane;
        const bool lands = (uint32_t)lane < ncan && fpos < ef;
        if (lands) F[fpos] = 1;
        EHX_GSYNC();
        for (uint32_t dhi = new_nR; dhi > p0;) {
          const uint32_t dlo = dhi - p0 > 64 ? dhi - 64 : p0;
          const uint32_t dpos = dlo + (uint32_t)lane;
          const bool in = dpos < dhi;
          const bool taken = in && F[dpos] != 0;
          const uint64_t occ = __ballot(taken);
          const uint32_t below = (uint32_t)__builtin_popcountll(__ballot(lands && fpos < dlo));
          const uint32_t cntb = below + (uint32_t)__builtin_popcountll(occ & lt_mask);
          const bool mv = in && !taken;
          uint64_t kj = 0;
          if (mv) kj = R[dpos - cntb];
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
      EHX_PROF(5)
    };
    if (nfresh != 0) {
      if (NCH == 1) {
        merge_keys(key[0], true);
      } else {
        // P = 4: a step holds up to 128 fresh keys, but only those below R's worst entry enter — in the steady state of
        // a search (R full) a handful.  When at most 64 enter they are compacted across the passes and merged ONCE
        // (rank, insertion points and the move of R are paid per merge); the early steps, which fill R, merge pass by pass.
        const uint64_t bound0 = nR < ef ? kKeyInf : R[ef - 1];
        uint64_t cm[NCH];
        uint32_t ntot = 0;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          cm[c] = __ballot(key[c] < bound0);
          ntot += (uint32_t)__builtin_popcountll(cm[c]);
        }
        if (ntot == 0) {
          request_next();
        } else if (ntot <= 64) {
          uint32_t off = 0;
#pragma unroll
          for (int c = 0; c < NCH; ++c) {
            if (key[c] < bound0) batch[off + (uint32_t)__builtin_popcountll(cm[c] & lt_mask)] = key[c];
            off += (uint32_t)__builtin_popcountll(cm[c]);
          }
          EHX_GSYNC();
          const uint64_t mk = (uint32_t)lane < ntot ? batch[lane] : kKeyInf;
          EHX_GSYNC();
          merge_keys(mk, true);
        } else {
#pragma unroll
          for (int c = 0; c < NCH; ++c)
            if ((uint32_t)c * 64u < nfresh) merge_keys(key[c], (uint32_t)(c + 1) * 64u >= nfresh);
        }
      }
    }
  }

  if (HELP) {  // release the helper wave
    if (lane == 0) ctrl[0] = 0xFFFFFFFFu;
    __syncthreads();
  }
  // ---- leave the visited bitmap all-zero (k_graph.hip) ----
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  if (a.vislog_cap == 0) {
  } else if (n_logged <= a.vislog_cap) {
    for (uint32_t i = lane; i < n_logged; i += 64) vis[vlog[i] >> 5] = 0u;
  } else {
    for (uint32_t i = lane; i < a.vis_words; i += 64) vis[i] = 0u;
  }
  // ---- results: the k closest of R ----
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
    atomicAdd(&a.counters[4], n_steps);
#ifdef EHX_GRAPH_PROFILE
    for (int i = 0; i < 7; ++i) atomicAdd(&a.counters[5 + i], prof_[i]);
#endif
  }
  if (a.done_flag) {
    __threadfence_system();
    __builtin_amdgcn_s_waitcnt(0);
    if (lane == 0) __hip_atomic_store(a.done_flag, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

hipError_t launch_graph_search_wide(const GraphArgs& a, hipStream_t st) {
  const uint32_t P = a.width >= 4 ? 4 : 2;
  const size_t lds = graph_lds_bytes(a.ld, a.ef_cap, P);
  static DynLdsAttr attr;
  const void* fns[8] = {(const void*)graph_search_wide_kernel<0, 2, false>, (const void*)graph_search_wide_kernel<1, 2, false>,
                        (const void*)graph_search_wide_kernel<0, 4, false>, (const void*)graph_search_wide_kernel<1, 4, false>,
                        (const void*)graph_search_wide_kernel<0, 2, true>,  (const void*)graph_search_wide_kernel<1, 2, true>,
                        (const void*)graph_search_wide_kernel<0, 4, true>,  (const void*)graph_search_wide_kernel<1, 4, true>};
  if (hipError_t e = attr.ensure(fns, 8, lds); e != hipSuccess) return e;
  const bool l2 = a.metric == 0;
  // The helper wave: rows of the lengths two share a 4-lane group (<= 256 dims: a pass of 32 rows per wave), batches of at most
  // one query per SIMD, and not the one-query form (its latency is launch + walk, not rows).
  // EHX_GRAPH_HELP=0 / 1 forces it off / on (A/B runs).
  const bool short_rows = a.dims <= 256 && (a.dims == 32 || a.dims == 64 || a.dims == 96 || a.dims == 128 || a.dims == 192 || a.dims == 256);
  // Measured (6.25 M x 128 L2, same box, profiles/r06_k_graph_6250k128_help{0,1}.jsonl): batch 1024, P = 4 — ef 200 0.474 ->
  // 0.528 of 8 TB/s, ef 800 0.519 -> 0.579, ef 50 0.383 -> 0.392; P = 2 — ef 200 0.440 -> 0.475.  Batch 2048 (two query waves per
  // SIMD already): 0.557 -> 0.490, so the helper is for batches of at most one query per SIMD.  2 M x 768 (forced on): +1 %.
  static std::atomic<int> n_simds{0};
  if (n_simds.load(std::memory_order_relaxed) == 0) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
      n_simds.store(cus * 4, std::memory_order_relaxed);
    else
      n_simds.store(1024, std::memory_order_relaxed);
  }
  const int help_env = env().graph_help;
  const bool help = help_env >= 0 ? help_env != 0 : (short_rows && (int)a.nq <= n_simds.load(std::memory_order_relaxed) && !a.q_raw);
#define EHX_LAUNCH_W(M, PP, H) \
  hipLaunchKernelGGL((graph_search_wide_kernel<M, PP, H>), dim3(a.nq), dim3(H ? 128 : 64), lds, st, a)
  if (help) {
    if (P == 2) { if (l2) EHX_LAUNCH_W(0, 2, true); else EHX_LAUNCH_W(1, 2, true); }
    else { if (l2) EHX_LAUNCH_W(0, 4, true); else EHX_LAUNCH_W(1, 4, true); }
  } else {
    if (P == 2) { if (l2) EHX_LAUNCH_W(0, 2, false); else EHX_LAUNCH_W(1, 2, false); }
    else { if (l2) EHX_LAUNCH_W(0, 4, false); else EHX_LAUNCH_W(1, 4, false); }
  }
#undef EHX_LAUNCH_W
  return hipGetLastError();
}

}  // namespace ehx
