#!/usr/bin/env python3
"""Graph-path measurement (SURVEY §8d graph rows): the oracle builds the HNSW graph on the host
(reference defaults M=16, efC=200, sequential), the engine imports it and serves batches of queries;
reports QPS, recall@k vs the engine's exact flat path, n_dist / n_hops per query, algorithmic bytes
per query and achieved HBM GB/s (= algorithmic bytes / kernel time), next to the oracle's CPU QPS on
all host cores at the same ef.  One JSON line per (dataset, ef)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=100_000)
    ap.add_argument("--dims", type=int, default=128)
    ap.add_argument("--metric", default="l2")
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--batches", default="",
                    help="comma list of batch sizes measured on the same index (default: --batch only): more queries "
                         "in flight = more waves per SIMD for the one-wave-per-query kernel")
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--efs", default="10,50,200")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--widths", default="1",
                    help="comma list of ehx_params.search_width values measured on the same index: 1 = the strict "
                         "(hnswlib-order) walk, 2 / 4 = the wide walk (k_graphw.hip)")
    ap.add_argument("--gpu-build", action="store_true", help="build the graph on the GPU (batched insertion); no oracle")
    ap.add_argument("--build-batch", type=int, default=0)
    ap.add_argument("--data", default="gauss",
                    help="gauss = EHX-GAUSS-1 (isotropic, the BASELINE workload: HNSW degenerates on it); "
                         "manifold:R = rows on an R-dimensional linear manifold (z ~ N(0, I_R) times a fixed random "
                         "R x dims matrix, plus 5 %% isotropic noise) — data with structure, where a graph index "
                         "reaches recall >= 0.95; generated on the host and written through ehx_set_batch")
    args = ap.parse_args()
    import torch  # noqa: F401  (same HIP runtime instance as the engine)
    import embeddinghub_amd as ehx
    from oracle import pyoracle
    em, om = {"l2": (ehx.METRIC_L2SQ, pyoracle.METRIC_L2), "cosine": (ehx.METRIC_COSINE, pyoracle.METRIC_COSINE),
              "ip": (ehx.METRIC_IP, pyoracle.METRIC_IP)}[args.metric]
    batches = [int(x) for x in args.batches.split(",")] if args.batches else [args.batch]
    n, d, B, k = args.rows, args.dims, max(batches), args.k
    norm = args.metric == "cosine"
    Q = pyoracle.gen_rows(ehx.SEED_QUERY, 0, B, d, normalize=norm)
    h = None
    manifold = None
    if args.data.startswith("manifold-dev:"):
        pass
    elif args.data.startswith("manifold:"):
        R = int(args.data.split(":")[1])
        A = np.random.default_rng(20250213).standard_normal((R, d)).astype(np.float32) / np.sqrt(R)

        def manifold(seed, rows):
            g = np.random.default_rng(seed)
            x = g.standard_normal((rows, R)).astype(np.float32) @ A
            x += 0.05 * g.standard_normal((rows, d)).astype(np.float32)
            if norm:
                x /= np.linalg.norm(x, axis=1, keepdims=True)
            return np.ascontiguousarray(x, dtype=np.float32)
        Q = manifold(ehx.SEED_QUERY + 1000, B)  # (+1000: SEED_QUERY + i equals the seed of corpus chunk i — queries drawn with it are noisy copies of corpus rows)
    if args.data.startswith("manifold-dev:"):
        # EHX-MANIFOLD-1 rows generated on the device (include/ehx_datagen.h), queries by the oracle's restatement of the
        # same generator under the query seed: what bench.py's graph_path_structured_10m leg runs
        R = int(args.data.split(":")[1])
        Q = pyoracle.gen_manifold_rows(ehx.SEED_QUERY, 0, B, d, R, normalize=norm)
        g = ehx.Space.unique("gbench", d, metric=em, mode=ehx.MODE_GRAPH, initial_capacity=n, build_batch=args.build_batch)
        t0 = time.perf_counter()
        g.fill_manifold(ehx.SEED_CORPUS, 0, n, R, norm)
        build_s = time.perf_counter() - t0
        flat = ehx.Space.unique("gbench-flat", d, metric=em, initial_capacity=n)
        flat.fill_manifold(ehx.SEED_CORPUS, 0, n, R, norm)
        builder = "GPU-built HNSW over EHX-MANIFOLD-1 rows, R = %d (batched insertion, %.0f rows/s)" % (R, n / build_s)
    elif manifold is not None:
        # host-generated rows through the public write path into a graph space (concurrent insertion rounds,
        # build_batch given explicitly) and a flat space (ground truth)
        g = ehx.Space.unique("gbench", d, metric=em, mode=ehx.MODE_GRAPH, initial_capacity=n,
                             build_batch=args.build_batch or 4096)
        flat = ehx.Space.unique("gbench-flat", d, metric=em, initial_capacity=n)
        build_s, chunk = 0.0, 65536
        for i0 in range(0, n, chunk):
            m = min(chunk, n - i0)
            X = manifold(ehx.SEED_CORPUS + 1 + i0 // chunk, m)
            keys = [b"%d" % i for i in range(i0, i0 + m)]
            t0 = time.perf_counter()
            g.set_batch(keys, X)
            build_s += time.perf_counter() - t0
            flat.set_batch(keys, X)
        builder = "GPU-built HNSW from ehx_set_batch (%s, rounds of <=%d rows, %.0f rows/s)" % (
            args.data, args.build_batch or 4096, n / build_s)
    elif args.gpu_build:
        g = ehx.Space.unique("gbench", d, metric=em, mode=ehx.MODE_GRAPH, initial_capacity=n, build_batch=args.build_batch)
        t0 = time.perf_counter()
        g.fill_synthetic(ehx.SEED_CORPUS, 0, n, norm)
        build_s = time.perf_counter() - t0
        builder = "GPU-built HNSW (batched insertion, %.0f rows/s)" % (n / build_s)
    else:
        X = pyoracle.gen_rows(ehx.SEED_CORPUS, 0, n, d, normalize=norm)
        h = pyoracle.Hnsw(d, om, n)
        build_s = h.add_rows(X)
        g = ehx.Space.unique("gbench", d, metric=em, mode=ehx.MODE_GRAPH, initial_capacity=n, build_batch=0xFFFFFFFF)
        g.fill_synthetic(ehx.SEED_CORPUS, 0, n, norm)
        l0, lv, upper = h.export_graph()
        g.graph_import(l0, lv, upper, h.enterpoint, h.maxlevel)
        builder = "oracle-built HNSW"
    if manifold is None and not args.data.startswith("manifold-dev:"):
        flat = ehx.Space.unique("gbench-flat", d, metric=em, initial_capacity=n)
        flat.fill_synthetic(ehx.SEED_CORPUS, 0, n, norm)
    truth, _, _ = flat.knn(Q, k)
    # the exact engine on the same data and queries, for the QPS-at-equal-recall comparison
    flat.stats_reset()
    t0 = time.perf_counter()
    for _ in range(3):
        flat.knn(Q, k)
    flat_qps = 3 * B / (time.perf_counter() - t0)
    cores = os.cpu_count() or 1
    Q_all, truth_all = Q, truth
    for B in batches:
        Q, truth = Q_all[:B], truth_all[:B]
        for ef, width in [(int(x), int(w)) for x in args.efs.split(",") for w in args.widths.split(",")]:
            g.set_ef(ef)
            g.set_search_width(width)
            if h is not None:
                h.set_ef(ef)
            ids, dist, cnt = g.knn(Q, k)  # warm-up
            g.stats_reset()
            t0 = time.perf_counter()
            for _ in range(args.reps):
                ids, dist, cnt = g.knn(Q, k)
            wall = (time.perf_counter() - t0) / args.reps
            st = g.stats()
            gc = g.graph_counters()
            recall = float(np.mean([len(set(ids[i]) & set(truth[i])) / k for i in range(B)]))
            best, same, orecall = 0.0, None, None
            if h is not None:
                labels, _, _, _, _ = h.search_batch(Q, k, threads=cores)
                for _ in range(3):
                    _, _, _, sec, ost = h.search_batch(Q, k, threads=cores)
                    best = max(best, B / sec)
                same = round(float(np.mean([np.array_equal(labels[i], ids[i]) for i in range(B)])), 4)
                orecall = round(float(np.mean([len(set(labels[i]) & set(truth[i])) / k for i in range(B)])), 4)
            bytes_q = st["bytes_algorithmic"] / (args.reps * B)
            kern_ms = st["scan_ms_mean"]
            print(json.dumps({
                "workload": "%dx%d %s, graph (%s, M=16 efC=200, build %.1fs), batch=%d k=%d ef=%d" % (
                    n, d, args.metric, builder, build_s, B, k, ef),
                "search_width": width, "expansions_per_step": round(gc[1] / gc[4], 3) if width > 1 and gc[4] else 1.0,
                "qps_host_pointers": round(B / wall, 1), "qps_kernel": round(B / (kern_ms * 1e-3), 1),
                "kernel_ms": round(kern_ms, 4), "recall_at_k": round(recall, 4), "oracle_recall_at_k": orecall,
                "queries_identical_to_oracle": same,
                "n_dist_per_query": round(st["n_dist"] / (args.reps * B), 1),
                "n_hops_per_query": round(st["n_hops"] / (args.reps * B), 1),
                "prefetch_hit_rate": round(gc[3] / max(gc[1], 1), 4),
                # -DEHX_GRAPH_PROFILE builds only: mean microseconds per level-0 expansion spent in
                # (pick next node, adjacency + visited, row fetch + distances, rank fresh keys, decide next + request,
                #  insertion points, move R, tail)
                "phase_us_per_hop": [round(v / 100.0 / max(gc[1], 1), 3) for v in gc[4:12]] if gc[4] and width <= 1 else None,
                # wide walk, -DEHX_GRAPH_PROFILE builds: microseconds per STEP spent in (pick, adjacency + visited +
                # compaction, row fetch + distances, rank + prediction, insertion points, move R, rest)
                "phase_us_per_step": [round(v / 100.0 / max(gc[4], 1), 3) for v in gc[5:12]] if width > 1 and gc[5] else None,
                "bytes_per_query": round(bytes_q, 1),
                "roofline": {"bound": "hbm", "achieved": round(bytes_q * B / (kern_ms * 1e-3) / 1e9, 2), "peak": 8000.0,
                             "unit": "GB/s", "frac": round(bytes_q * B / (kern_ms * 1e-3) / 8e12, 5)},
                "cpu_oracle_qps": round(best, 1) if h is not None else None, "cpu_cores": cores,
                "flat_exact_qps_host_pointers": round(flat_qps, 1),
            }), flush=True)

if __name__ == "__main__":
    main()
