// Graph-mode insertion on the GPU: hnswlib's addPoint (call site
// embeddinghub/embeddingstore/index.cc:36; algorithm restated in oracle/hnsw_oracle.hpp:
// addPoint / searchBaseLayer / getNeighborsByHeuristic2 / mutuallyConnectNewElement) for a batch of
// P new rows whose vectors are already in HBM.
//
//   insert_search_kernel — one wavefront per new node: greedy descent through the levels above the
//     node's level, then per level L..0 a best-first search bounded by ef_construction over the
//     graph AS IT IS BEFORE THE BATCH, followed by hnswlib's neighbour-selection heuristic; emits the
//     selected neighbours per level (<= M each) and uses the closest one as the next level's entry.
//   insert_link_kernel — one wavefront per (existing node, level) that gained new neighbours:
//     appends the new ids while the list has room, otherwise re-selects with the heuristic over
//     {new id} u old list exactly as mutuallyConnectNewElement does; also writes the new nodes' own
//     lists (farthest first, hnswlib's pop order).
//
// With P = 1 this is hnswlib's sequential insertion: the graph is identical to the oracle's
// whenever the distances that get compared are distinct (ties are ordered by id here and by heap
// layout in libstdc++).  With P > 1 it is the analogue of hnswlib's multi-threaded add_items: new
// nodes of one batch do not see each other, the result depends on the batch size, and parity is
// recall parity, not graph identity.
//
// All distances use the canonical (oracle-order) arithmetic and read the SEARCH COPY only (cosine rows normalised
// there: exactly the vectors hnswlib would have stored) — the raw rows may be fp32 or binary16.
#include "ehx_kernels.h"

namespace ehx {

#ifndef EHX_I_WSYNC
#define EHX_I_WSYNC 1  // every kernel here runs one wave per workgroup: wave_lds_sync() instead of a barrier
#endif
#if EHX_I_WSYNC
#define EHX_ISYNC() wave_lds_sync()
#else
#define EHX_ISYNC() __syncthreads()
#endif

namespace {

constexpr uint32_t kNone = 0xFFFFFFFFu;

__device__ __forceinline__ uint64_t wsort64(uint64_t key, int lane) {
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

__device__ __forceinline__ uint32_t lb_lds(const uint64_t* a, uint32_t n, uint64_t key) {
  uint32_t lo = 0, hi = n;
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (a[mid] < key) lo = mid + 1;
    else hi = mid;
  }
  return lo;
}

// canonical distance between two STORED rows, one lane, both read from the SEARCH COPY (k_misc.hip: cosine rows
// already normalised — the product hnswlib stores — and every 16-float block permuted so that piece j holds the four
// inputs of SSE partial sum j in order).  Piece j of a block therefore feeds partial sum j with its four products one
// after the other: the same additions in the same order as walking the raw rows 16 bytes at a time, and the kernels
// here need neither the raw rows nor their norms (fp16 row storage: the search copy is made from the rounded rows).
// sa / sb: per-row scales applied to the elements on the fly (single-copy graph spaces, cosine: the rows are stored
// raw; x * inv_norm is the normalised row hnswlib-python stores, one rounding per element).  1.0f: the rows as stored
// (a multiplication by one is exact).
__device__ __forceinline__ float row_row_dist(int metric01, const float* __restrict__ xa, const float* __restrict__ xb,
                                              uint32_t dims, float sa = 1.0f, float sb = 1.0f) {
  uint32_t body;
  if ((dims & 15u) == 0 || (dims & 3u) == 0) body = dims;
  else if (dims > 16) body = dims & ~15u;
  else if (dims > 4) body = dims & ~3u;
  else body = 0;
  float p[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  auto step = [&](float4 a, float4 b, float& acc, int ncomp) {
    a.x = ex_mul(a.x, sa); a.y = ex_mul(a.y, sa); a.z = ex_mul(a.z, sa); a.w = ex_mul(a.w, sa);
    b.x = ex_mul(b.x, sb); b.y = ex_mul(b.y, sb); b.z = ex_mul(b.z, sb); b.w = ex_mul(b.w, sb);
    if (metric01 == 0) {
      const float d0 = ex_sub(a.x, b.x), d1 = ex_sub(a.y, b.y), d2 = ex_sub(a.z, b.z), d3 = ex_sub(a.w, b.w);
      acc = ex_add(acc, ex_mul(d0, d0));
      if (ncomp > 1) acc = ex_add(acc, ex_mul(d1, d1));
      if (ncomp > 2) acc = ex_add(acc, ex_mul(d2, d2));
      if (ncomp > 3) acc = ex_add(acc, ex_mul(d3, d3));
    } else {
      acc = ex_add(acc, ex_mul(a.x, b.x));
      if (ncomp > 1) acc = ex_add(acc, ex_mul(a.y, b.y));
      if (ncomp > 2) acc = ex_add(acc, ex_mul(a.z, b.z));
      if (ncomp > 3) acc = ex_add(acc, ex_mul(a.w, b.w));
    }
  };
  const float4* a4 = (const float4*)xa;
  const float4* b4 = (const float4*)xb;
  const uint32_t n16 = body / 16u;
  // two blocks (sixteen 16-byte pieces of both rows) requested before the first is used: a one-piece-per-trip
  // loop keeps ONE load in flight and pays the cache latency dims/4 times
  uint32_t t = 0;
  for (; t + 2 <= n16; t += 2) {
    float4 ra[8], rb[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      ra[i] = a4[t * 4 + i];
      rb[i] = b4[t * 4 + i];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) step(ra[i], rb[i], p[i & 3], 4);
  }
  for (; t < n16; ++t) {
#pragma unroll
    for (int j = 0; j < 4; ++j) step(a4[t * 4 + j], b4[t * 4 + j], p[j], 4);
  }
  const int rem4 = (int)((body & 15u) >> 2);  // 4-float pieces of a last, partial block: components 0..rem4-1
  if (rem4) {
#pragma unroll
    for (int j = 0; j < 4; ++j) step(a4[n16 * 4 + j], b4[n16 * 4 + j], p[j], rem4);
  }
  float res = ex_add(ex_add(ex_add(p[0], p[1]), p[2]), p[3]);
  if (body != dims) {
    float tail = 0.0f;
    for (uint32_t m = body; m < dims; ++m) {
      const uint32_t pos = search_copy_pos(m);
      const float va = ex_mul(xa[pos], sa), vb = ex_mul(xb[pos], sb);
      if (metric01 == 0) {
        const float d = ex_sub(va, vb);
        tail = ex_add(tail, ex_mul(d, d));
      } else {
        tail = ex_add(tail, ex_mul(va, vb));
      }
    }
    if (metric01 != 0 && body) return ex_sub(ex_add(ex_sub(1.0f, res), ex_sub(1.0f, tail)), 1.0f);  // see canon_dist
    res = body ? ex_add(res, tail) : tail;
  }
  if (metric01 != 0) res = ex_sub(1.0f, res);
  return res;
}

// hnswlib getNeighborsByHeuristic2 on one wave.  cand[0..nc): (dist-to-base, id<<1|flag) keys sorted
// ascending; if nc < Msel all are kept.  Otherwise candidates are visited closest first and kept iff
// no already-kept r has dist(r, c) < dist(c, base).  kept ids/keys go to kept[0..nk) in visiting
// order (closest first).  Returns nk (uniform).
__device__ __forceinline__ uint32_t select_heuristic(const InsertArgs& a, const uint64_t* cand, uint32_t nc,
                                                     uint32_t Msel, uint64_t* kept, int lane) {
  const int metric01 = a.metric == 0 ? 0 : 1;
  if (nc < Msel) {
    for (uint32_t i = lane; i < nc; i += 64) kept[i] = cand[i];
    EHX_ISYNC();
    return nc;
  }
  uint32_t nk = 0;
  for (uint32_t i = 0; i < nc && nk < Msel; ++i) {
    const uint64_t ck = cand[i];
    const uint32_t cid = (uint32_t)(ck & 0xFFFFFFFFull) >> 1;
    const float dq = ordered_to_f32((uint32_t)(ck >> 32));
    bool bad = false;
    if ((uint32_t)lane < nk) {
      const uint32_t rid = (uint32_t)(kept[lane] & 0xFFFFFFFFull) >> 1;
      const float d = row_row_dist(metric01, a.Xs + (size_t)rid * a.ld, a.Xs + (size_t)cid * a.ld, a.dims,
                                   a.xscale ? a.xscale[rid] : 1.0f, a.xscale ? a.xscale[cid] : 1.0f);
      bad = d < dq;
    }
    if (!__any(bad)) {
      if (lane == 0) kept[nk] = ck;
      nk += 1;
    }
    EHX_ISYNC();
  }
  return nk;
}

}  // namespace

// LDS per wave: q[ld] | R[ef] | R2[ef] | batch[64] | kept[64] (u64) | ids[64] (u32)
size_t insert_lds_bytes(uint32_t ld, uint32_t ef) { return (size_t)ld * 4 + (size_t)ef * 16 + 64 * 8 * 2 + 64 * 4; }

__global__ __launch_bounds__(64) void insert_search_kernel(const InsertArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x;
  const uint32_t p = blockIdx.x;
  float* qs = (float*)smem;
  uint64_t* R = (uint64_t*)(smem + (size_t)a.ld * 4);
  uint64_t* R2 = R + a.ef;
  uint64_t* batch = R2 + a.ef;
  uint64_t* kept = batch + 64;
  uint32_t* ids_l = (uint32_t*)(kept + 64);
  uint32_t* vis = a.visited + (size_t)p * a.vis_words;
  uint32_t* vlog = a.vislog + (size_t)p * a.vislog_cap;

  const uint32_t me = a.new_ids ? a.new_ids[p] : a.id0 + p;
  const int my_level = a.new_levels[p];
  const int metric01 = a.metric == 0 ? 0 : 1;
  // the query is the new row itself, prepared the way hnswlib stores it: its search-copy row (normalised for
  // cosine, permuted like every row the distance passes read)
  {
    const float qsc = a.xscale ? a.xscale[me] : 1.0f;   // (single-copy graph spaces: the stored row is raw)
    for (uint32_t i = lane; i < a.ld; i += 64) qs[i] = ex_mul(a.Xs[(size_t)me * a.ld + i], qsc);
  }
  EHX_ISYNC();

  // canonical distances of rows ids_l[0..count) to the new row: 4-lane groups reading the search copy in coalesced
  // 64-byte pieces (wave_group_dists, as k_graph.hip); lane p gets row p
  auto lane_dist = [&](uint32_t count) -> float {
    return metric01 == 0 ? wave_group_dists<0>(qs, a.Xs, a.ld, a.dims, ids_l, count, lane)
                         : wave_group_dists<1>(qs, a.Xs, a.ld, a.dims, ids_l, count, lane, a.xscale);
  };
  auto list_of = [&](uint32_t node, int level, uint32_t* width) -> const uint32_t* {
    if (level == 0) {
      *width = a.M0;
      return a.adj0 + (size_t)node * a.M0;
    }
    *width = a.M;
    return a.up_lists + ((size_t)a.up_start[node] + (uint32_t)(level - 1)) * a.M;
  };

  uint32_t* out = a.sel ? a.sel + (size_t)p * (a.max_sel_levels * (1 + a.M)) : nullptr;
  if (out) {
    for (uint32_t i = lane; i < a.max_sel_levels * (1 + a.M); i += 64) out[i] = (i % (1 + a.M)) == 0 ? 0u : kNone;
    __syncthreads();  // (global memory handed between lanes — here: rewritten by other lanes later — keeps the real fence)
  }

  // ---- greedy descent through the levels above the node's level (hnswlib addPoint) ----
  uint32_t cur = a.entry_point;
  if (lane == 0) ids_l[0] = cur;
  EHX_ISYNC();
  float curdist = 0.0f;
  if (my_level < a.max_level) {
    curdist = wave_uniform(lane_dist(1));
    for (int level = a.max_level; level > my_level; --level) {
      bool changed = true;
      while (changed) {
        changed = false;
        uint32_t width;
        const uint32_t* lst = list_of(cur, level, &width);
        uint32_t nb = kNone;
        if (lane < (int)width) nb = lst[lane];
        const uint32_t cnt = __builtin_popcountll(__ballot(nb != kNone));
        if (lane < (int)cnt) ids_l[lane] = nb;
        EHX_ISYNC();
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
        m = wave_uniform(m);  // (the butterfly leaves the minimum in every lane)
        mi = wave_uniform(mi);
        if (m < curdist) {
          curdist = m;
          cur = wave_uniform(ids_l[mi]);
          changed = true;
        }
        EHX_ISYNC();
      }
    }
  }

  // ---- per level: searchBaseLayer(ef_construction) + heuristic(M) ----
  const uint32_t ef = a.ef;
  const int top_level = my_level < a.max_level ? my_level : a.max_level;
  for (int level = top_level; level >= 0; --level) {
    uint32_t nlog = 0;
    if (lane == 0) ids_l[0] = cur;
    EHX_ISYNC();
    const float d0 = wave_uniform(lane_dist(1));
    uint32_t nR = 1;
    if (lane == 0) {
      R[0] = ((uint64_t)f32_to_ordered(d0) << 32) | ((uint64_t)cur << 1);
      atomicOr(&vis[cur >> 5], 1u << (cur & 31));
      vlog[0] = cur;
    }
    nlog = 1;
    EHX_ISYNC();
    for (;;) {
      uint32_t idx = kNone;
      for (uint32_t base = 0; base < nR && idx == kNone; base += 64) {
        const uint32_t i = base + lane;
        const bool un = i < nR && !(R[i] & 1ull);
        const uint64_t m = __ballot(un);
        if (m) idx = base + (uint32_t)__builtin_ctzll(m);
      }
      if (idx == kNone) break;
      const uint32_t c = wave_uniform((uint32_t)(R[idx] & 0xFFFFFFFFull) >> 1);
      EHX_ISYNC();
      if (lane == 0) R[idx] |= 1ull;
      uint32_t width;
      const uint32_t* lst = list_of(c, level, &width);
      uint32_t nb = kNone;
      if (lane < (int)width) nb = lst[lane];
      bool fresh = false;
      if (nb != kNone) {
        const uint32_t bit = 1u << (nb & 31);
        fresh = !(atomicOr(&vis[nb >> 5], bit) & bit);
      }
      const uint64_t fmask = __ballot(fresh);
      const uint32_t nfresh = __builtin_popcountll(fmask);
      const uint32_t slot = __builtin_popcountll(fmask & ((1ull << lane) - 1ull));
      if (fresh) {
        ids_l[slot] = nb;
        if (nlog + slot < a.vislog_cap) vlog[nlog + slot] = nb;
      }
      nlog += nfresh;
      EHX_ISYNC();
      if (nfresh == 0) continue;
      uint64_t mykey = kKeyInf;
      {
        const float d = lane_dist(nfresh);
        if ((uint32_t)lane < nfresh) mykey = ((uint64_t)f32_to_ordered(d) << 32) | ((uint64_t)ids_l[lane] << 1);
      }
      mykey = wsort64(mykey, lane);
      batch[lane] = mykey;
      EHX_ISYNC();
      if ((uint32_t)lane < nfresh) {
        const uint32_t pos = lb_lds(R, nR, mykey) + lane;
        if (pos < ef) R2[pos] = mykey;
      }
      for (uint32_t j = lane; j < nR; j += 64) {
        const uint64_t kj = R[j];
        const uint32_t pos = j + lb_lds(batch, nfresh, kj);
        if (pos < ef) R2[pos] = kj;
      }
      EHX_ISYNC();
      nR = nR + nfresh < ef ? nR + nfresh : ef;
      uint64_t* t = R;
      R = R2;
      R2 = t;
    }
    // reset the visited bits of this search (hnswlib takes a fresh visited tag per searchBaseLayer)
    __syncthreads();  // the log was written by other lanes, through global memory
    if (nlog <= a.vislog_cap) {
      for (uint32_t i = lane; i < nlog; i += 64) {
        const uint32_t v = vlog[i];
        atomicAnd(&vis[v >> 5], ~(1u << (v & 31)));
      }
    } else {
      for (uint32_t i = lane; i < a.vis_words; i += 64) vis[i] = 0u;
    }
    EHX_ISYNC();
    if (a.exclude_self) {
      // repairConnectionsForUpdate: the node being updated is part of the graph and finds itself;
      // hnswlib filters it out of the results (and skips the level if nothing else is left)
      uint32_t pos = kNone;
      for (uint32_t base = 0; base < nR && pos == kNone; base += 64) {
        const uint32_t i = base + lane;
        const uint64_t m = __ballot(i < nR && ((uint32_t)(R[i] & 0xFFFFFFFFull) >> 1) == me);
        if (m) pos = base + (uint32_t)__builtin_ctzll(m);
      }
      if (pos != kNone) {
        for (uint32_t i = lane; i < nR; i += 64) R2[i] = R[i];
        EHX_ISYNC();
        for (uint32_t i = pos + lane; i + 1 < nR; i += 64) R[i] = R2[i + 1];
        EHX_ISYNC();
        nR -= 1;
      }
      if (nR == 0) continue;  // level skipped: own list and entry for the next level unchanged
    }
    // heuristic over the (sorted) results; selection is always with M, even at level 0
    const uint32_t nk = select_heuristic(a, R, nR, a.M, kept, lane);
    // own list: farthest first (hnswlib pops the max-heap); next entry = the closest selected
    if (out) {
      uint32_t* o = out + (size_t)level * (1 + a.M);
      if (lane == 0) o[0] = nk;
      if ((uint32_t)lane < nk) o[1 + lane] = (uint32_t)(kept[nk - 1 - lane] & 0xFFFFFFFFull) >> 1;
    }
    if (a.link_head) {
      // bulk build: the node writes its own list here (nothing reaches a node of this round before the link kernel
      // has run: the graph the searches walk is the graph before the round) and registers one (list, new node) pair
      // per selected neighbour — a linked list per adjacency list, heads in link_head, so the link kernel needs no
      // host-side regrouping.  The order pairs arrive in is arbitrary; the link kernel applies them by ascending id.
      uint32_t width;
      uint32_t* own = const_cast<uint32_t*>(list_of(me, level, &width));
      const uint32_t t = (uint32_t)lane < nk ? (uint32_t)(kept[nk - 1 - lane] & 0xFFFFFFFFull) >> 1 : kNone;
      if ((uint32_t)lane < width) own[lane] = t;
      if (t != kNone) {
        const uint32_t lid = level == 0 ? t : a.head_rows + a.up_start[t] + (uint32_t)(level - 1);
        const uint32_t pair = (p * a.max_sel_levels + (uint32_t)level) * a.M + (uint32_t)lane;
        const uint32_t old = atomicExch(&a.link_head[lid], pair + 1u);
        a.link_next[pair] = old;
        if (old == 0u) a.link_touched[atomicAdd(a.link_count, 1u)] = make_uint2(t, (uint32_t)level);
      }
    }
    cur = (uint32_t)(kept[0] & 0xFFFFFFFFull) >> 1;
    EHX_ISYNC();
  }
}

hipError_t launch_insert_search(const InsertArgs& a, uint32_t n_new, hipStream_t st) {
  const size_t lds = insert_lds_bytes(a.ld, a.ef);
  static DynLdsAttr attr;
  const void* fns[1] = {(const void*)insert_search_kernel};
  if (hipError_t e = attr.ensure(fns, 1, lds); e != hipSuccess) return e;
  hipLaunchKernelGGL(insert_search_kernel, dim3(n_new), dim3(64), lds, st, a);
  return hipGetLastError();
}

// mutuallyConnectNewElement for ONE incoming id on one wave: the list `lst` (width entries, kNone-padded) of node s
// receives nid — appended while there is room, otherwise re-selected with the heuristic over {nid} u list.
// is_update: hnswlib's isUpdate branch (a link that already exists is left alone).
__device__ __forceinline__ void link_incoming(const InsertArgs& a, uint32_t* lst, uint32_t width, uint32_t s,
                                              uint32_t nid, bool is_update, uint64_t* cand, uint64_t* kept, int lane,
                                              int metric01) {
  uint32_t nb = kNone;
  if ((uint32_t)lane < width) nb = lst[lane];
  const uint32_t cnt = __builtin_popcountll(__ballot(nb != kNone));
  if (is_update && __any(nb == nid)) return;  // isUpdate: the link already exists
  if (cnt < width) {
    if (lane == 0) lst[cnt] = nid;
    __syncthreads();
    return;
  }
  // full: candidates = {new} u list with distances to s, heuristic with the level's max degree
  auto key_of = [&](uint32_t id) {
    const float d = row_row_dist(metric01, a.Xs + (size_t)id * a.ld, a.Xs + (size_t)s * a.ld, a.dims,
                                 a.xscale ? a.xscale[id] : 1.0f, a.xscale ? a.xscale[s] : 1.0f);
    return ((uint64_t)f32_to_ordered(d) << 32) | ((uint64_t)id << 1);
  };
  if (cnt < 64) {
    const uint32_t id = (uint32_t)lane < cnt ? nb : ((uint32_t)lane == cnt ? nid : kNone);
    uint64_t key = kKeyInf;
    if (id != kNone) key = key_of(id);
    key = wsort64(key, lane);
    cand[lane] = key;
  } else {
    // 64 list entries fill the wave: sort them, then slot the incoming key in at its rank (keys are distinct:
    // the id is part of the key and the incoming id is not in the list)
    uint64_t nkey = 0;
    if (lane == 0) nkey = key_of(nid);
    nkey = ((uint64_t)__shfl((int)(nkey >> 32), 0, 64) << 32) | (uint32_t)__shfl((int)(uint32_t)nkey, 0, 64);
    const uint64_t key = wsort64(key_of(nb), lane);
    const uint32_t rank = __builtin_popcountll(__ballot(key < nkey));
    cand[(uint32_t)lane + ((uint32_t)lane >= rank ? 1u : 0u)] = key;
    if (lane == 0) cand[rank] = nkey;
  }
  __syncthreads();
  const uint32_t nk = select_heuristic(a, cand, cnt + 1, width, kept, lane);
  // rewrite farthest first
  if ((uint32_t)lane < width) lst[lane] = (uint32_t)lane < nk ? (uint32_t)(kept[nk - 1 - lane] & 0xFFFFFFFFull) >> 1 : kNone;
  __syncthreads();
}

// One wave per work item w: target node tgt[w] at level tlevel[w] receives the new ids
// inc_ids[inc_off[w] .. inc_off[w+1]) in that order.  kind[w] == 1: the target IS a new node and the
// ids are its own selected list (already farthest first) — plain overwrite.  (Host-built items: single-row
// Sets of an existing key, graph_update.)
__global__ __launch_bounds__(64) void insert_link_kernel(const InsertArgs a, const uint32_t* __restrict__ tgt,
                                                         const int32_t* __restrict__ tlevel,
                                                         const uint32_t* __restrict__ kind,
                                                         const uint32_t* __restrict__ inc_off,
                                                         const uint32_t* __restrict__ inc_ids) {
  __shared__ uint64_t cand[65];  // a full level-0 list of M = 32 holds 64 ids; the incoming one is the 65th candidate
  __shared__ uint64_t kept[64];
  const int lane = threadIdx.x;
  const uint32_t w = blockIdx.x;
  const uint32_t s = tgt[w];
  const int level = tlevel[w];
  const uint32_t width = level == 0 ? a.M0 : a.M;
  uint32_t* lst = level == 0 ? a.adj0 + (size_t)s * a.M0
                             : a.up_lists + ((size_t)a.up_start[s] + (uint32_t)(level - 1)) * a.M;
  const uint32_t b0 = inc_off[w], b1 = inc_off[w + 1];
  if (kind[w] == 1) {
    if ((uint32_t)lane < width) lst[lane] = (b0 + lane < b1) ? inc_ids[b0 + lane] : kNone;
    return;
  }
  const int metric01 = a.metric == 0 ? 0 : 1;
  for (uint32_t b = b0; b < b1; ++b) link_incoming(a, lst, width, s, inc_ids[b], kind[w] == 2, cand, kept, lane, metric01);
}

// Bulk build: the link work items come from the search kernel's registrations.  Waves stride over the touched lists;
// a wave gathers the list's chain of pairs (new node = id0 + pair / (max_sel_levels * M)), clears the head for the
// next round, and applies the new ids in ASCENDING id order — the order of insertion, whatever order the pairs were
// registered in: the graph after the round is a function of the searches' results only.
constexpr uint32_t kIncCap = 1024;  // incoming ids of one list kept in LDS; a longer chain is re-walked per id
__global__ __launch_bounds__(64) void insert_link_dev_kernel(const InsertArgs a) {
  __shared__ uint64_t cand[65];
  __shared__ uint64_t kept[64];
  __shared__ uint32_t inc[kIncCap];
  __shared__ uint32_t chain_len;
  const int lane = threadIdx.x;
  const int metric01 = a.metric == 0 ? 0 : 1;
  const uint32_t n_items = *(volatile const uint32_t*)a.link_count;
  const uint32_t per_node = a.max_sel_levels * a.M;
  for (uint32_t w = blockIdx.x; w < n_items; w += gridDim.x) {
    const uint2 tl = a.link_touched[w];
    const uint32_t s = tl.x;
    const int level = (int)tl.y;
    const uint32_t width = level == 0 ? a.M0 : a.M;
    uint32_t* lst = level == 0 ? a.adj0 + (size_t)s * a.M0
                               : a.up_lists + ((size_t)a.up_start[s] + (uint32_t)(level - 1)) * a.M;
    const uint32_t lid = level == 0 ? s : a.head_rows + a.up_start[s] + (uint32_t)(level - 1);
    const uint32_t h0 = a.link_head[lid];
    if (lane == 0) {
      uint32_t n = 0;
      for (uint32_t pr = h0; pr != 0u; pr = a.link_next[pr - 1u]) {
        if (n < kIncCap) inc[n] = a.id0 + (pr - 1u) / per_node;
        n += 1;
      }
      chain_len = n;
      a.link_head[lid] = 0u;
    }
    __syncthreads();
    const uint32_t c = chain_len;
    uint32_t prev = 0;
    for (uint32_t b = 0; b < c; ++b) {
      // the smallest incoming id above the last one applied
      uint32_t m = kNone;
      if (c <= kIncCap) {
        for (uint32_t i = lane; i < c; i += 64) {
          const uint32_t v = inc[i];
          if ((b == 0 || v > prev) && v < m) m = v;
        }
      } else if (lane == 0) {
        for (uint32_t pr = h0; pr != 0u; pr = a.link_next[pr - 1u]) {
          const uint32_t v = a.id0 + (pr - 1u) / per_node;
          if ((b == 0 || v > prev) && v < m) m = v;
        }
      }
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const uint32_t other = (uint32_t)__shfl_xor((int)m, o, 64);
        m = other < m ? other : m;
      }
      m = wave_uniform(m);
      prev = m;
      link_incoming(a, lst, width, s, m, false, cand, kept, lane, metric01);
    }
    __syncthreads();
  }
}

// hnswlib updatePoint, first half: every one-hop neighbour `neigh` of the updated node re-selects its
// links among sCand \ {neigh} (sCand = the updated node, its one-hop and two-hop neighbours): keep the
// min(ef_construction, |candidates|) closest to neigh, run the heuristic with the level's max degree,
// rewrite the list farthest first.  One wave per neighbour; candidates are passed by the host.
__global__ __launch_bounds__(64) void update_neigh_kernel(const InsertArgs a, const uint32_t* __restrict__ neigh,
                                                          int level, const uint32_t* __restrict__ cand_off,
                                                          const uint32_t* __restrict__ cand_ids) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint64_t* R = (uint64_t*)smem;
  uint64_t* R2 = R + a.ef;
  uint64_t* batch = R2 + a.ef;
  uint64_t* kept = batch + 64;
  const int lane = threadIdx.x;
  const uint32_t w = blockIdx.x;
  const uint32_t s = neigh[w];
  const uint32_t c0 = cand_off[w], c1 = cand_off[w + 1];
  const uint32_t width = level == 0 ? a.M0 : a.M;
  uint32_t* lst = level == 0 ? a.adj0 + (size_t)s * a.M0
                             : a.up_lists + ((size_t)a.up_start[s] + (uint32_t)(level - 1)) * a.M;
  const int metric01 = a.metric == 0 ? 0 : 1;
  const uint32_t keep = (c1 - c0) < a.ef ? (c1 - c0) : a.ef;
  uint32_t nR = 0;
  for (uint32_t base = c0; base < c1; base += 64) {
    const uint32_t n_here = c1 - base < 64 ? c1 - base : 64;
    uint64_t key = kKeyInf;
    if ((uint32_t)lane < n_here) {
      const uint32_t id = cand_ids[base + lane];
      const float d = row_row_dist(metric01, a.Xs + (size_t)s * a.ld, a.Xs + (size_t)id * a.ld, a.dims,
                                   a.xscale ? a.xscale[s] : 1.0f, a.xscale ? a.xscale[id] : 1.0f);
      key = ((uint64_t)f32_to_ordered(d) << 32) | ((uint64_t)id << 1);
    }
    key = wsort64(key, lane);
    batch[lane] = key;
    EHX_ISYNC();
    if ((uint32_t)lane < n_here) {
      const uint32_t pos = lb_lds(R, nR, key) + lane;
      if (pos < keep) R2[pos] = key;
    }
    for (uint32_t j = lane; j < nR; j += 64) {
      const uint64_t kj = R[j];
      const uint32_t pos = j + lb_lds(batch, n_here, kj);
      if (pos < keep) R2[pos] = kj;
    }
    EHX_ISYNC();
    nR = nR + n_here < keep ? nR + n_here : keep;
    uint64_t* t = R;
    R = R2;
    R2 = t;
  }
  const uint32_t nk = select_heuristic(a, R, nR, width, kept, lane);
  if ((uint32_t)lane < width) lst[lane] = (uint32_t)lane < nk ? (uint32_t)(kept[nk - 1 - lane] & 0xFFFFFFFFull) >> 1 : kNone;
}

hipError_t launch_update_neigh(const InsertArgs& a, uint32_t n_items, const uint32_t* neigh, int level,
                               const uint32_t* cand_off, const uint32_t* cand_ids, hipStream_t st) {
  if (n_items == 0) return hipSuccess;
  const size_t lds = (size_t)a.ef * 16 + 64 * 8 * 2;
  hipLaunchKernelGGL(update_neigh_kernel, dim3(n_items), dim3(64), lds, st, a, neigh, level, cand_off, cand_ids);
  return hipGetLastError();
}

hipError_t launch_insert_link(const InsertArgs& a, uint32_t n_items, const uint32_t* tgt, const int32_t* tlevel,
                              const uint32_t* kind, const uint32_t* inc_off, const uint32_t* inc_ids, hipStream_t st) {
  if (n_items == 0) return hipSuccess;
  hipLaunchKernelGGL(insert_link_kernel, dim3(n_items), dim3(64), 0, st, a, tgt, tlevel, kind, inc_off, inc_ids);
  return hipGetLastError();
}

hipError_t launch_insert_link_dev(const InsertArgs& a, uint32_t n_waves, hipStream_t st) {
  if (n_waves == 0) return hipSuccess;
  hipLaunchKernelGGL(insert_link_dev_kernel, dim3(n_waves), dim3(64), 0, st, a);
  return hipGetLastError();
}

}  // namespace ehx
