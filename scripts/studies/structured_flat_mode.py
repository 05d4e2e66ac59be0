#!/usr/bin/env python3
"""Round 5: where do the 6 ms per batch go in the exact engine's slow mode on structured rows?  (gpu_r05_a: the scan
phase is 0.82-0.85 ms in BOTH modes, no fallbacks, no adaptation — the difference is outside the scan events.)

The structured leg of bench.py without the graph index: the flat space over the manifold rows, ten batches through
ehx_knn_device, wall time per batch beside the engine's own event times (scan phase, first event -> last event), on
the NULL stream (what bench.py's structured leg passes: torch's default stream) and on a stream of its own; optionally
with a graph space alive beside it (--graph-rows).  One JSON line per phase."""
import argparse
import json
import sys
import time
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import embeddinghub_amd as ehx  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=1_000_000)
ap.add_argument("--dims", type=int, default=768)
ap.add_argument("--R", type=int, default=16)
ap.add_argument("--batches", type=int, default=10)
ap.add_argument("--graph-rows", type=int, default=0)
ap.add_argument("--gauss", action="store_true", help="isotropic rows instead (control)")
ap.add_argument("--label", default="")
ap.add_argument("--own-first", action="store_true", help="first phase on the stream of its own instead of the NULL stream")
ap.add_argument("--device-queries", action="store_true", help="query batches generated on the device (no pageable upload)")
ap.add_argument("--sleep", type=float, default=0.0, help="seconds between the end of the build and the first search")
args = ap.parse_args()
n2, d, R, B, k, chunk = args.rows, args.dims, args.R, 1024, 10, 65536
A = np.random.default_rng(20250213).standard_normal((R, d)).astype(np.float32) / np.sqrt(R)


def manifold(seed, rows):
    r = np.random.default_rng(seed)
    if args.gauss:
        x = r.standard_normal((rows, d)).astype(np.float32)
    else:
        x = r.standard_normal((rows, R)).astype(np.float32) @ A
        x += 0.05 * r.standard_normal((rows, d)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    return np.ascontiguousarray(x, dtype=np.float32)


if os.environ.get("SFM_BLAS_THREADS"):   # cap numpy's BLAS pool (default: one spinning thread per visible core)
    from threadpoolctl import threadpool_limits
    threadpool_limits(limits=int(os.environ["SFM_BLAS_THREADS"]))


flat = ehx.Space.unique("sfm-flat", d, metric=ehx.METRIC_COSINE, initial_capacity=n2)
g = None
if args.graph_rows:
    g = ehx.Space.unique("sfm-graph", d, metric=ehx.METRIC_COSINE, mode=ehx.MODE_GRAPH, initial_capacity=args.graph_rows,
                         build_batch=4096)
for i0 in range(0, n2, chunk):
    m = min(chunk, n2 - i0)
    X = manifold(ehx.SEED_CORPUS + 1 + i0 // chunk, m)
    keys = [b"%d" % i for i in range(i0, i0 + m)]
    if g is not None and i0 < args.graph_rows:
        g.set_batch(keys[:max(0, min(m, args.graph_rows - i0))], X[:max(0, min(m, args.graph_rows - i0))])
    flat.set_batch(keys, X)
nb = args.batches
sq = torch.empty((nb, B, d), dtype=torch.float32, device="cuda")
if args.device_queries:
    import ctypes as C
    from embeddinghub_amd import _lib
    for i in range(nb):
        _lib.check(_lib.load().ehx_gen_rows_device(C.c_void_p(0), ehx.SEED_QUERY, i * B, B, d, 1, C.c_void_p(sq[i].data_ptr())))
else:
    for i in range(nb):
        sq[i].copy_(torch.from_numpy(manifold(ehx.SEED_QUERY + 1000 + i, B)))
torch.cuda.synchronize()
ti = torch.empty((B, k), dtype=torch.int64, device="cuda")
td = torch.empty((B, k), dtype=torch.float32, device="cuda")
tc = torch.empty((B,), dtype=torch.int32, device="cuda")
own = torch.cuda.Stream()


def throttled():
    try:
        kv = dict(l.split()[:2] for l in open("/sys/fs/cgroup/cpu.stat"))
        return int(kv["nr_throttled"]), int(kv["throttled_usec"])
    except (OSError, KeyError, ValueError):
        return None


def phase(name, stream_handle, sync):
    thr0 = throttled()
    for _ in range(3):
        flat.knn_device(sq[0], k, ti, td, tc, stream=stream_handle)
        sync()
    flat.stats_reset()
    per = []
    t00 = time.perf_counter()
    for i in range(nb):
        t0 = time.perf_counter()
        flat.knn_device(sq[i], k, ti, td, tc, stream=stream_handle)
        t1 = time.perf_counter()
        sync()
        per.append((round((t1 - t0) * 1e3, 3), round((time.perf_counter() - t1) * 1e3, 3)))
    wall = time.perf_counter() - t00
    st = flat.stats()
    thr1 = throttled()
    print(json.dumps({"cgroup_throttled_times_ms": None if thr0 is None or thr1 is None else [thr1[0] - thr0[0], round((thr1[1] - thr0[1]) / 1e3, 1)],
                      "label": args.label, "phase": name, "qps": round(nb * B / wall, 1), "wall_ms_per_batch": round(wall / nb * 1e3, 3),
                      "scan_ms_mean": round(st["scan_ms_mean"], 4), "last_scan_ms": round(st["last_scan_ms"], 4),
                      "last_total_ms": round(st["last_total_ms"], 4), "fallbacks": int(st["n_i8_fallback"] + st["n_filter_fallback"]),
                      "call_ms_then_sync_ms": per[:6], "slowest_call": max(range(nb), key=lambda i: per[i][0] + per[i][1]),
                      "slowest_call_ms_then_sync_ms": max(per, key=lambda p: p[0] + p[1]), "ids_checksum": int(ti.sum().item())}), flush=True)


if args.sleep:
    time.sleep(args.sleep)
if args.own_first:
    phase("own stream, stream sync (first)", own.cuda_stream, own.synchronize)
phase("null stream, device sync", torch.cuda.current_stream().cuda_stream, torch.cuda.synchronize)
phase("own stream, stream sync", own.cuda_stream, own.synchronize)
phase("null stream, device sync (again)", torch.cuda.current_stream().cuda_stream, torch.cuda.synchronize)
