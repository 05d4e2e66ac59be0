"""ORACLE — TEST INFRASTRUCTURE ONLY.  CPU model of the engine's WIDE graph walk (ehx_params.search_width = 2 / 4,
embeddinghub_amd/csrc/k_graphw.hip), in plain Python, for small cases.

Parity unpinned against the reference: the reference has no such mode — its only walk is hnswlib's searchKnn /
searchBaseLayerST (call site embeddinghub/embeddingstore/index.cc:41; restated in oracle/hnsw_oracle.hpp), one
candidate per step.  This model states the ONE deviation the wide mode makes from that algorithm and nothing else, so
that the kernel can be checked bit for bit against something that is not itself:

  * upper levels: hnswlib's greedy descent, unchanged (first strictly-closest neighbour in stored order);
  * level 0: hnswlib keeps a candidate heap, a result heap bounded by ef and a lower bound; equivalently ONE list R of at
    most ef (distance, id) entries, each expanded or not, where the next node is the closest unexpanded entry and the
    search ends when none is left (argument in k_graph.hip's header).  The wide walk takes the P closest unexpanded
    entries of R per step instead of one: their neighbour lists are read in list order, every neighbour not yet visited
    is marked and evaluated, and R <- the ef smallest of R u {fresh}.  Same distances (the oracle's canonical fp32
    arithmetic, passed in), same (distance, id) order, same termination.

The result is a deterministic function of (graph, distances, ef, P): which of two lists names a shared neighbour first
does not matter, the fresh SET is the same.  Counters as SURVEY §8d: n_dist = rows evaluated, n_hops0 = nodes expanded.
"""
import numpy as np


def _key(d, i):
    return (float(d), int(i))


def wide_search(level0, upper, entry_point, max_level, dist_row, ef, P, k):
    """level0: [n, 1 + 2M] u32 rows (count, ids...) as Hnsw.export_graph() returns them; upper: {(node, level): ids};
    dist_row[i] = the oracle's canonical distance of the query to row i (float32); returns (ids, dists, counters)."""
    n_dist = n_hops0 = n_hops_up = steps = 0
    cur = int(entry_point)
    curdist = np.float32(dist_row[cur])
    n_dist += 1
    for level in range(int(max_level), 0, -1):
        changed = True
        while changed:
            changed = False
            lst = upper.get((cur, level), ())
            n_hops_up += 1
            n_dist += len(lst)
            best, best_d = None, curdist
            for nb in lst:                       # first strictly-smaller minimum in stored order
                d = np.float32(dist_row[int(nb)])
                if d < best_d:
                    best, best_d = int(nb), d
            if best is not None:
                # (hnswlib moves to each improving neighbour as it scans; the engine takes the list's minimum — the same
                #  node ends the scan either way, see k_graph.hip)
                cur, curdist, changed = best, best_d, True
    ef = max(int(ef), int(k))
    R = [(_key(curdist, cur), False)]            # sorted by key; flag = expanded
    visited = {cur}
    while True:
        picks = [j for j, (_, done) in enumerate(R) if not done][:P]
        if not picks:
            break
        steps += 1
        nodes = []
        for j in picks:
            R[j] = (R[j][0], True)
            nodes.append(R[j][0][1])
        fresh = []
        for c in nodes:
            n_hops0 += 1
            row = level0[c]
            for nb in row[1:1 + int(row[0])]:
                nb = int(nb)
                if nb not in visited:
                    visited.add(nb)
                    fresh.append(nb)
        n_dist += len(fresh)
        R.extend((_key(np.float32(dist_row[i]), i), False) for i in fresh)
        R.sort(key=lambda e: e[0])
        del R[ef:]
    top = R[:k]
    ids = np.array([e[0][1] for e in top], dtype=np.uint64)
    dists = np.array([e[0][0] for e in top], dtype=np.float32)
    return ids, dists, {"n_dist": n_dist, "n_hops0": n_hops0, "n_hops_up": n_hops_up, "steps": steps}
