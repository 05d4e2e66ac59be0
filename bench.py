#!/usr/bin/env python3
"""bench.py — headline benchmark of the kNN hot path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch: 1024 cosine queries (k=10) against the
10M x 768 fp32 corpus of EHX-GAUSS-1 (include/ehx_datagen.h), resident in HBM.  With N > 1
(launched by torch.distributed.run, one rank per GPU over RCCL) the corpus is row-sharded —
rank r holds rows [r*N/G, (r+1)*N/G) — every rank scans its shard for the same query batch,
the local top-k lists are all-gathered over xGMI and merged on every rank (strong scaling:
the total index is fixed).

Prints ONE JSON line on rank 0 (contract in the task statement), with `roofline` for the scan
kernel (matrix-core bound) and `cpu_baseline` (the oracle = restated hnswlib, on the host cores,
on a bounded sample of the same workload; N=1 only).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
MFMA_F16_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_f16, dense (no sparsity)
METRIC_NAME = "kNN queries/sec at recall@10>=0.95, 10Mx768 cosine, batch=1024"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--rows", type=int, default=10_000_000, help="total corpus rows (all GPUs)")
    ap.add_argument("--dims", type=int, default=768)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--rows-dtype", choices=["f32", "f16"], default="f32",
                    help="row storage in HBM (f16 = BASELINE configs[5] storage; arithmetic stays fp32)")
    ap.add_argument("--scan", choices=["auto", "f32"], default="auto",
                    help="scan engine of the timed run: auto = fp16 matrix-core filter + certified fp32 re-rank "
                         "(default), f32 = fp32 matrix-core scan only; results are identical")
    ap.add_argument("--metric-kind", choices=["cosine", "l2", "ip"], default="cosine",
                    help="distance of the timed space (the headline metric is cosine; l2/ip: other BASELINE configs)")
    ap.add_argument("--set-stream", type=int, default=0, metavar="ROWS",
                    help="also time ehx_set_batch of ROWS fresh rows from host memory into a second space of the same "
                         "shape (BASELINE configs[4]: streamed Set); reported as set_stream")
    ap.add_argument("--set-concurrent", type=int, default=0, metavar="ROWS",
                    help="BASELINE configs[4] as written — streamed Set CONCURRENT with GetNeighbors: a writer thread "
                         "streams ROWS fresh rows (ehx_set_batch, chunks of 8192) into a space of the bench shape (up to "
                         "1 M rows, explicit keys) while the main thread keeps searching it; both rates reported as "
                         "set_concurrent (rank 0)")
    ap.add_argument("--no-f32-engine", action="store_true", help="skip the short A/B leg on the fp32-only engine")
    ap.add_argument("--graph-rows", type=int, default=1_000_000,
                    help="N=1 only: also measure the graph path (GPU-built HNSW, M=16 efC=200) over the first ROWS corpus "
                         "rows with the same queries — reported as graph_path with its HBM roofline; 0 = skip")
    ap.add_argument("--graph-ef", type=int, default=100)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-rows", type=int, default=16000)
    ap.add_argument("--cpu-sample-queries", type=int, default=256)
    return ap.parse_args()


def cpu_baseline(args):
    """Oracle (restated hnswlib, SSE order) on the host cores, bounded sample of the workload:
    sequential HNSW build (M=16, efC=200, seed=100 — the reference's defaults) over the first
    S corpus rows, then batched search at the smallest ef reaching recall@10 >= 0.95 against the
    oracle's own exhaustive search over the same S rows."""
    import numpy as np
    from oracle import pyoracle
    cores = os.cpu_count() or 1
    S, nq, d, k = args.cpu_sample_rows, args.cpu_sample_queries, args.dims, args.k
    X = pyoracle.gen_rows(20250211, 0, S, d, normalize=True)
    Q = pyoracle.gen_rows(20250212, 0, nq, d, normalize=True)
    truth, _, _, ex_sec = pyoracle.exhaustive(X, Q, k, pyoracle.METRIC_COSINE, threads=cores, return_time=True)
    h = pyoracle.Hnsw(d, pyoracle.METRIC_COSINE, S)
    build_sec = h.add_rows(X)
    best = None
    for ef in (40, 160, 640, 1280, 2560, 5120, 10240):
        h.set_ef(ef)
        labels, _, _, sec, st = h.search_batch(Q, k, threads=cores)
        recall = float(np.mean([len(set(labels[i]) & set(truth[i])) / k for i in range(nq)]))
        best = dict(ef=ef, recall=recall, qps=nq / sec, n_dist=st["n_dist"] / nq, n_hops=(st["n_hops0"] + st["n_hops_up"]) / nq)
        if recall >= 0.95:
            break
    h.set_ef(best["ef"])
    for _ in range(3):  # best of 3: the first multi-threaded pass after the serial build runs cold
        _, _, _, sec, _ = h.search_batch(Q, k, threads=cores)
        best["qps"] = max(best["qps"], nq / sec)
    for _ in range(2):
        _, _, _, ex2 = pyoracle.exhaustive(X, Q, k, pyoracle.METRIC_COSINE, threads=cores, return_time=True)
        ex_sec = min(ex_sec, ex2)
    _, _, _, sec1, _ = h.search_batch(Q[:64], k, threads=1)
    return {
        "value": round(best["qps"], 1), "unit": "queries/s", "cores": cores, "kind": "port",
        "sample": ("oracle HNSW (M=16, efC=200, sequential build %.1fs) over the first %d of %d corpus rows, %d queries, "
                   "ef=%d -> recall@10=%.3f, n_dist/query=%.0f; single-thread %.1f q/s; oracle exhaustive scan of the "
                   "same sample %.1f q/s on %d threads (= %.2e row-distances/s)" % (
                       build_sec, S, args.rows, nq, best["ef"], best["recall"], best["n_dist"], 64 / sec1,
                       nq / ex_sec, cores, nq * S / ex_sec)),
        "ef": best["ef"], "recall_at_10": round(best["recall"], 4),
        "exhaustive_row_dists_per_s": nq * S / ex_sec,
    }


def graph_path(args, ehx, torch, queries, metric, stream):
    """The graph path on the same workload shape: HNSW (reference defaults M=16, efC=200) built on the GPU over the
    first --graph-rows corpus rows, batches of the same queries through ehx_knn_device; HBM roofline from the
    kernel's own work counters (rows fetched x row bytes + adjacency rows) and HIP-event kernel time;
    recall against the exact flat engine over the same rows."""
    n, d, B, k, ef = min(args.graph_rows, args.rows), args.dims, args.batch, args.k, args.graph_ef
    norm = True  # the same (unit-normalised) corpus rows as the timed space
    g = ehx.Space("bench-graph", d, metric=metric, mode=ehx.MODE_GRAPH, initial_capacity=n)
    t0 = time.perf_counter()
    g.fill_synthetic(ehx.SEED_CORPUS, 0, n, norm)
    build_s = time.perf_counter() - t0
    g.set_ef(ef)
    ids = torch.empty((B, k), dtype=torch.int64, device="cuda")
    dst = torch.empty((B, k), dtype=torch.float32, device="cuda")
    cnt = torch.empty((B,), dtype=torch.int32, device="cuda")
    reps = min(5, queries.shape[0])
    g.knn_device(queries[0], k, ids, dst, cnt, stream=stream)  # warm-up
    torch.cuda.synchronize()
    g.stats_reset()
    t0 = time.perf_counter()
    for i in range(reps):
        g.knn_device(queries[i], k, ids, dst, cnt, stream=stream)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / reps
    st = g.stats()
    got = ids.cpu().numpy()
    flat = ehx.Space("bench-graph-truth", d, metric=metric, initial_capacity=n)
    flat.fill_synthetic(ehx.SEED_CORPUS, 0, n, norm)
    flat.knn_device(queries[reps - 1], k, ids, dst, cnt, stream=stream)
    torch.cuda.synchronize()
    truth = ids.cpu().numpy()
    recall = float(sum(len(set(got[i]) & set(truth[i])) for i in range(B))) / (B * k)
    flat.drop()
    g.drop()
    kern_ms = st["scan_ms_mean"]
    bytes_q = st["bytes_algorithmic"] / (reps * B)
    ach = bytes_q * B / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
    return {
        "workload": "%dx%d %s (EHX-GAUSS-1), GPU-built HNSW M=16 efC=200 (%.0f rows/s), batch=%d, k=%d, ef=%d" % (
            n, d, args.metric_kind, n / build_s, B, k, ef),
        "value": round(B / wall, 1), "unit": "queries/s", "ms_per_step": round(wall * 1e3, 4),
        "recall_at_10": round(recall, 4),
        "note": "isotropic Gaussian rows: no graph index reaches recall 0.95 here (DESIGN.md); the exact scan is the "
                "headline path, this leg reports the graph kernel's HBM roofline on the same workload shape",
        "rows_fetched_per_query": round(st["n_dist"] / (reps * B), 1),
        "expansions_per_query": round(st["n_hops"] / (reps * B), 1),
        "roofline": {"bound": "hbm", "kernel": "graph_search_kernel", "achieved": round(ach, 1), "peak": 8000.0,
                     "unit": "GB/s", "frac": round(ach / 8000.0, 4), "traffic": None, "kernel_ms": round(kern_ms, 4),
                     "algorithmic_bytes_per_query": round(bytes_q, 1)},
    }


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    import embeddinghub_amd as ehx
    from embeddinghub_amd import _lib
    L = _lib.load()
    dev = (C.c_int * 1)(local_rank)
    _lib.check(L.ehx_init(dev, 1))

    from embeddinghub_amd.sharded import shard_range
    G = world
    row0, shard = shard_range(args.rows, G, rank)
    B, d, k = args.batch, args.dims, args.k
    t_fill = time.time()
    metric = {"cosine": ehx.METRIC_COSINE, "l2": ehx.METRIC_L2SQ, "ip": ehx.METRIC_IP}[args.metric_kind]
    space = ehx.Space("bench-r%d" % rank, d, metric=metric,
                      initial_capacity=shard,
                      dtype=ehx.DTYPE_F16 if args.rows_dtype == "f16" else ehx.DTYPE_F32)
    space.fill_synthetic(ehx.SEED_CORPUS, row0, shard, True)
    torch.cuda.synchronize()
    t_fill = time.time() - t_fill

    stream = torch.cuda.current_stream().cuda_stream
    n_batches = args.warmup + args.steps
    queries = torch.empty((n_batches, B, d), dtype=torch.float32, device="cuda")
    for i in range(n_batches):  # distinct query batches, resident in HBM before the timed region
        _lib.check(L.ehx_gen_rows_device(C.c_void_p(stream), ehx.SEED_QUERY, i * B, B, d, 1,
                                         C.c_void_p(queries[i].data_ptr())))
    from embeddinghub_amd.sharded import ShardedSearcher
    searcher = ShardedSearcher(row0, B, k, "cuda", space=space, stream=stream)

    def step(i):
        searcher.knn(queries[i])

    def barrier():
        if G > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(steps, warmup):
        """warmup untimed steps, then exactly `steps` timed ones; returns (seconds, stats)"""
        for i in range(warmup):
            step(i % n_batches)
        barrier()
        space.stats_reset()
        t0 = time.perf_counter()
        for i in range(steps):
            step((warmup + i) % n_batches)
        torch.cuda.synchronize()
        if G > 1:
            dist.barrier()
            torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if G > 1:
            t = torch.tensor([el], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, space.stats()  # stats: per-batch scan-phase durations from HIP events on the launch stream

    f16_rows = args.rows_dtype == "f16"
    filt = args.scan == "auto" and os.environ.get("EHX_SCAN", "") != "f32"
    if not filt:
        space.set_scan(ehx.SCAN_F32)
    elapsed, st = timed(args.steps, args.warmup)
    scan_ms = st["scan_ms_mean"]
    flops_per_batch = 2.0 * B * shard * d                    # 2*B*N*d per batch (this shard)
    achieved = flops_per_batch / (scan_ms * 1e-3) / 1e12 if scan_ms > 0 else 0.0
    peak = MFMA_F16_PEAK_TFLOPS if filt else MFMA_F32_PEAK_TFLOPS
    row_bytes = 2 if (filt or f16_rows) else 4                # bytes per element the scan streams from HBM
    algo_bytes = shard * d * row_bytes + B * d * 4 + B * k * 12
    traffic = None  # measured HBM bytes per batch (PMC pass, gfx950-corrected) when this exact workload was profiled
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            key = "%s:%dx%d:b%d:g%d" % ("filter" if filt else "f32", args.rows, d, B, G)
            traffic = json.load(f).get(key)
    except (OSError, ValueError):
        pass
    engine = ("fp16 matrix-core filter scan (lower-bound scores, k'=k+22 candidates) + canonical fp32 re-rank with "
              "per-query certification; uncertified queries re-scanned in fp32" if filt else
              "fp32 matrix-core scan + canonical fp32 re-rank")
    out = {
        "metric": METRIC_NAME, "value": round(args.steps * B / elapsed, 1), "unit": "queries/s",
        "n_gpus": G, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f16 filter -> f32 exact results" if filt else "f32", "data": "synthetic",
        "config": {
            "workload": "%dx%d %s (EHX-GAUSS-1 seed %d), batch=%d, k=%d; exhaustive scan = exact kNN "
                        "(recall@10 = 1.0 vs exhaustive by construction; ids and distances bit-identical to the "
                        "fp32 oracle)" % (args.rows, d, args.metric_kind, ehx.SEED_CORPUS, B, k),
            "engine": engine,
            "rows_total": args.rows, "rows_per_gpu": shard, "rows_dtype": args.rows_dtype, "dims": d, "batch": B,
            "k": k, "path": "flat",
            "parallelism": "row-shard x%d + all-gather top-k merge" % G if G > 1 else "single GPU",
            "fill_seconds": round(t_fill, 2),
        },
        "recall_at_10": 1.0,
        "roofline": {
            "bound": "mfma",
            "kernel": ("flat_scan16_kernel: the cascade of scan passes of one batch (launches + inter-pass merges)"
                       if filt else "flat_scan8_kernel: the scan passes of one batch"),
            "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
            "traffic": traffic,
            "kernel_ms": round(scan_ms, 4), "batches_timed": int(st["scan_launches"]),
            "flops_per_batch": flops_per_batch, "algorithmic_bytes_per_batch": algo_bytes,
            "hbm_frac_of_8TBps": round(algo_bytes / (scan_ms * 1e-3) / 8e12, 4) if scan_ms > 0 else None,
            "note": ("fp16 MFMA on real data is power-limited on this part: scripts/ubench/mfma_f16_data.hip "
                     "sustains 1.27-1.65 PFLOP/s on unit-vector data vs 2.46 on constants "
                     "(profiles/r01_g_mfma_data_ubench.txt)") if filt else None,
        },
        "n_uncertified": int(st["n_uncertified"]),
        "filter_fallback_queries": int(st["n_filter_fallback"]),
    }
    if filt and not args.no_f32_engine:
        # A/B leg: the same space, same queries, fp32-only engine (identical results by construction)
        space.set_scan(ehx.SCAN_F32)
        el2, st2 = timed(2, 1)
        ms2 = st2["scan_ms_mean"]
        ach2 = flops_per_batch / (ms2 * 1e-3) / 1e12 if ms2 > 0 else 0.0
        out["f32_scan_engine"] = {
            "value": round(2 * B / el2, 1), "unit": "queries/s", "ms_per_step": round(el2 / 2 * 1e3, 3),
            "kernel_ms": round(ms2, 4), "achieved": round(ach2, 2), "peak": MFMA_F32_PEAK_TFLOPS,
            "frac": round(ach2 / MFMA_F32_PEAK_TFLOPS, 4), "n_uncertified": int(st2["n_uncertified"]),
        }
        space.set_scan(ehx.SCAN_AUTO)
    if args.graph_rows and G == 1:
        out["graph_path"] = graph_path(args, ehx, torch, queries, metric, stream)
    if args.set_stream and rank == 0:
        # streamed Set: host rows -> engine (pinned staging, H2D, per-row statistics, scan copy), while nothing
        # else runs; rows/s as a cgo caller of ehx_set_batch would see it
        import numpy as np
        m = args.set_stream
        rows = np.random.default_rng(1).standard_normal((m, d)).astype(np.float32)
        keys = ["s%d" % i for i in range(m)]
        w = ehx.Space("bench-set-r%d" % rank, d, metric=metric, initial_capacity=m,
                      dtype=ehx.DTYPE_F16 if args.rows_dtype == "f16" else ehx.DTYPE_F32)
        chunk = 65536
        t0 = time.perf_counter()
        for i0 in range(0, m, chunk):
            w.set_batch(keys[i0:i0 + chunk], rows[i0:i0 + chunk])
        dt = time.perf_counter() - t0
        out["set_stream"] = {"rows": m, "rows_per_s": round(m / dt, 1), "GB_per_s_host": round(m * d * 4 / dt / 1e9, 3),
                             "chunk_rows": chunk}
        w.drop()
    if args.set_concurrent and rank == 0 and G == 1:
        # streamed Set concurrent with search on the SAME space (explicit keys, so its own space: up to 1 M rows of
        # the bench shape written through ehx_set_batch first — that load is the un-contended Set rate): writers
        # take the space's write lock per chunk (row upload, per-row statistics, scan copy), searches take it
        # shared per batch — every batch sees a consistent row count; both sides timed over the writer's lifetime
        import threading
        import numpy as np
        m, chunk = args.set_concurrent, 8192
        base = min(shard, 1_000_000)
        rng = np.random.default_rng(2)

        def gen(nrows):
            x = rng.standard_normal((nrows, d), dtype=np.float32)
            x /= np.linalg.norm(x, axis=1, keepdims=True)
            return x
        w = ehx.Space("bench-conc", d, metric=metric, initial_capacity=base + m,
                      dtype=ehx.DTYPE_F16 if args.rows_dtype == "f16" else ehx.DTYPE_F32)
        load_s = 0.0
        for i0 in range(0, base, 65536):
            x = gen(min(65536, base - i0))
            ks = ["b%d" % i for i in range(i0, i0 + x.shape[0])]
            t0 = time.perf_counter()
            w.set_batch(ks, x)
            load_s += time.perf_counter() - t0
        rows = gen(m)
        keys = ["c%d" % i for i in range(m)]
        wsearch = ShardedSearcher(0, B, k, "cuda", space=w, stream=stream)
        wsearch.knn(queries[0])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(5):
            wsearch.knn(queries[i % n_batches])
        torch.cuda.synchronize()
        alone_qps = 5 * B / (time.perf_counter() - t0)
        state = {"dt": None, "err": None}

        def writer():
            try:
                t0 = time.perf_counter()
                for i0 in range(0, m, chunk):
                    w.set_batch(keys[i0:i0 + chunk], rows[i0:i0 + chunk])
                state["dt"] = time.perf_counter() - t0
            except Exception as e:  # noqa: BLE001 - reported in the JSON line
                state["err"] = repr(e)

        th = threading.Thread(target=writer)
        t0 = time.perf_counter()
        th.start()
        batches = 0
        while th.is_alive():
            wsearch.knn(queries[batches % n_batches])
            torch.cuda.synchronize()
            batches += 1
        th.join()
        el = time.perf_counter() - t0
        out["set_concurrent"] = {
            "space": "%d x %d %s rows written through ehx_set_batch (%.0f rows/s un-contended), then %d more while "
                     "searching" % (base, d, args.rows_dtype, base / load_s, m),
            "rows_written_meanwhile": len(w) - base,
            "set_rows_per_s_meanwhile": round(m / state["dt"], 1) if state["dt"] else None,
            "search_queries_per_s_meanwhile": round(batches * B / el, 1), "search_batches_meanwhile": batches,
            "search_queries_per_s_alone": round(alone_qps, 1), "chunk_rows": chunk, "error": state["err"],
        }
        w.drop()
    if rank == 0 and G == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if G > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
