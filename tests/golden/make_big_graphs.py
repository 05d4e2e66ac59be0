#!/usr/bin/env python3
"""Oracle-built HNSW graphs at scale, for tests/test_graph_scale.py.

The oracle (oracle/hnsw_oracle.hpp: hnswlib restated, sequential addPoint, reference defaults M=16 / efC=200 /
seed=100) builds ~1-5 k rows/s on one core — minutes for these sizes — so the graphs are generated OFFLINE by this
script and kept out of the history (tests/golden/_big/ is git-ignored; it travels to the GPU box with the repo
snapshot like the built .so files).  The rows are not stored: they are EHX-GAUSS-1 (include/ehx_datagen.h), which the
engine regenerates on the device bit-for-bit (tests/test_datagen.py).  Stored per configuration: the graph (level-0
lists in stored order, levels, upper lists, entry point), the oracle's own search results (ids, distance bytes, counts,
work counters) for the first 256 EHX-GAUSS-1 queries at ef = 10 / 100 / 400, the oracle's exhaustive top-10 of those
queries, and the build time with the core count.

    python tests/golden/make_big_graphs.py [name ...]      (names: see CONFIGS; default: all)
"""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import pyoracle  # noqa: E402

SEED_CORPUS, SEED_QUERY = 20250211, 20250212
CONFIGS = {
    # name: (rows, dims, metric, normalise)
    "cos200k768": (200_000, 768, pyoracle.METRIC_COSINE, True),     # BASELINE configs[2] shape (cosine, d = 768)
    "l2_1m128": (1_000_000, 128, pyoracle.METRIC_L2, False),        # BASELINE configs[3] shape (L2, d = 128)
    "cos20k768": (20_000, 768, pyoracle.METRIC_COSINE, True),       # small twin of the first (quick local runs)
    # Rows on a 32-dim linear manifold + 5 % noise, with the QUERIES OF ROUND 3's structured leg: their generator seed
    # (SEED_QUERY + i) equals a corpus chunk's (SEED_CORPUS + 1 + i), so every query is a corpus row's latent point under
    # fresh noise — a near-duplicate lookup (nearest neighbour at cosine distance 0.003), far easier for a graph than
    # independent queries (0.31).  Found at the end of round 3; bench.py's leg now draws independent queries (below).
    # These two stay for what they are good for: "man200k768" is the structured index of tests/test_graph_scale.py
    # (imported-graph identity and GPU-built vs oracle-built recall do not depend on how the queries were drawn);
    # "manifold1m768" is the CPU number that belongs to the GPU numbers measured on those queries in round 3.
    "manifold1m768": (1_000_000, 768, pyoracle.METRIC_COSINE, "manifold"),
    "man200k768": (200_000, 768, pyoracle.METRIC_COSINE, "manifold"),
    # bench.py's structured leg as it is now: rows on a 16-dim linear manifold + 5 % noise, INDEPENDENT queries (seeds
    # SEED_QUERY + 1000 + i).  CPU numbers only (profiles/r03_cpu_hnsw_*.json), no graph kept.
    "s16_1m768": (1_000_000, 768, pyoracle.METRIC_COSINE, "manifold:16"),
    "s16_200k768": (200_000, 768, pyoracle.METRIC_COSINE, "manifold:16"),
}
NO_GRAPH_FILE = ("manifold1m768", "s16_1m768")   # (round 4: s16_200k768 keeps its graph — tests/test_graph_scale.py's
                                                  # structured index with INDEPENDENT queries, recall 0.965 at ef = 40)
EFS = (10, 100, 400)
NQ, K = 256, 10


def make(name):
    n, d, metric, norm = CONFIGS[name]
    out = os.path.join(HERE, "_big", name + ".npz")
    cores = os.cpu_count()
    t0 = time.time()
    if isinstance(norm, str):  # bench.py's run_structured_leg rows and queries ("manifold": round 3's, see CONFIGS)
        R, chunk = (int(norm.split(":")[1]) if ":" in norm else 32), 65536
        q_seed = SEED_QUERY + 1000 if ":" in norm else SEED_QUERY
        A = np.random.default_rng(20250213).standard_normal((R, d)).astype(np.float32) / np.sqrt(R)

        def manifold(seed, rows):
            r = np.random.default_rng(seed)
            x = r.standard_normal((rows, R)).astype(np.float32) @ A
            x += 0.05 * r.standard_normal((rows, d)).astype(np.float32)
            x /= np.linalg.norm(x, axis=1, keepdims=True)
            return np.ascontiguousarray(x, dtype=np.float32)
        X = np.concatenate([manifold(SEED_CORPUS + 1 + i0 // chunk, min(chunk, n - i0)) for i0 in range(0, n, chunk)])
        Q = manifold(q_seed, 1024)[:NQ]  # (bench.py's first query batch)
    else:
        X = pyoracle.gen_rows(SEED_CORPUS, 0, n, d, normalize=norm)
        Q = pyoracle.gen_rows(SEED_QUERY, 0, NQ, d, normalize=norm)
    print("[%s] rows generated in %.1f s" % (name, time.time() - t0), flush=True)
    import hashlib
    rows_sha1 = hashlib.sha1(X.tobytes()).hexdigest()  # (the structured rows come out of a float32 matmul: the test checks
    h = pyoracle.Hnsw(d, metric, n)                     # that its host produced the same bytes before it demands identity)
    t0 = time.time()
    step = max(1, n // 20)
    for r0 in range(0, n, step):
        h.add_rows(X[r0:r0 + step], first_label=r0)
        print("[%s] %d rows inserted, %.0f s" % (name, min(n, r0 + step), time.time() - t0), flush=True)
    build_s = time.time() - t0
    l0, lv, upper = h.export_graph()
    items = sorted(upper.items())
    un = np.array([k[0] for k, _ in items], dtype=np.uint32)
    ul = np.array([k[1] for k, _ in items], dtype=np.int32)
    off = np.zeros(len(items) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(v) for _, v in items])
    uids = np.concatenate([np.asarray(v, dtype=np.uint32) for _, v in items]) if items else np.zeros(0, np.uint32)
    truth, tdist, _ = pyoracle.exhaustive(X, Q, K, metric)
    res = {}
    meta = {"name": name, "rows": n, "dims": d, "metric": int(metric), "normalize": norm if isinstance(norm, str) else bool(norm), "M": 16,
            "ef_construction": 200, "seed": 100, "build_seconds": round(build_s, 1), "build_threads": 1,
            "host_cores": cores, "queries": NQ, "k": K, "rows_sha1": rows_sha1, "search": {}}
    efs = (10, 20, 40, 100, 200) if isinstance(norm, str) else EFS  # (the structured leg's range)
    meta["efs"] = list(efs)
    for ef in efs:
        h.set_ef(ef)
        labels, dists, counts, sec, st = h.search_batch(Q, K, threads=1)
        res["ids_ef%d" % ef] = labels.astype(np.uint64)
        res["dist_ef%d" % ef] = dists
        res["cnt_ef%d" % ef] = counts
        recall = float(np.mean([len(set(labels[i].tolist()) & set(truth[i].tolist())) / K for i in range(NQ)]))
        t1 = time.time()
        h.search_batch(Q, K, threads=cores)
        qps_all = NQ / (time.time() - t1)
        meta["search"][str(ef)] = dict(st, recall_at_10=round(recall, 5), qps_1_thread=round(NQ / sec, 1) if sec else None,
                                       qps_all_cores=round(qps_all, 1))
        print("[%s] ef=%d recall@10 %.4f n_dist/query %.0f" % (name, ef, recall, st["n_dist"] / NQ), flush=True)
    prof = os.path.join(ROOT, "profiles", "r03_cpu_hnsw_%s.json" % name)
    if os.path.exists(prof) and os.environ.get("EHX_KEEP_PROFILE"):  # a re-run that only needs the graph file
        prof = os.devnull
    with open(prof, "w") as f:  # the CPU baseline of the graph path, like for like (bench.py quotes these files)
        json.dump(meta, f, indent=1)
    if name in NO_GRAPH_FILE:
        return
    np.savez_compressed(out, level0=l0, levels=lv, upper_node=un, upper_level=ul, upper_off=off, upper_ids=uids,
                        entry_point=np.uint32(h.enterpoint), max_level=np.int32(h.maxlevel), truth=truth.astype(np.uint64),
                        truth_dist=tdist, meta=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8), **res)
    print("[%s] wrote %s (%.1f MB), oracle build %.0f s" % (name, out, os.path.getsize(out) / 1e6, build_s), flush=True)


if __name__ == "__main__":
    for nm in (sys.argv[1:] or list(CONFIGS)):
        make(nm)
