#!/usr/bin/env python3
"""BASELINE configs[0] as the reference runs it: 10 000 x 128 L2, ONE query per GetNeighbors call, ef = 10 (the reference's
default) and ef = 200 — through the C ABI (ehx_knn from host memory, one call per query, graph mode with the
sequentially built = oracle-identical graph, and flat mode), next to the CPU oracle on one thread (the reference server is
single-threaded under its mutex, server.cc:175).  Latency per call, not throughput: this is the plumbing configuration."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401
import embeddinghub_amd as ehx  # noqa: E402
from oracle import pyoracle  # noqa: E402

n, d, k, nq = 10_000, 128, 10, 500
X = pyoracle.gen_rows(ehx.SEED_CORPUS, 0, n, d, normalize=False)
Q = pyoracle.gen_rows(ehx.SEED_QUERY, 0, nq, d, normalize=False)
truth, _, _ = pyoracle.exhaustive(X, Q, k, pyoracle.METRIC_L2)
h = pyoracle.Hnsw(d, pyoracle.METRIC_L2, n)
t0 = time.perf_counter()
h.add_rows(X)
oracle_build_s = time.perf_counter() - t0
g = ehx.Space.unique("c1-graph", d, metric=ehx.METRIC_L2SQ, mode=ehx.MODE_GRAPH, initial_capacity=n)
t0 = time.perf_counter()
g.set_batch([b"%d" % i for i in range(n)], X)          # sequential insertion on the GPU: the oracle's graph
gpu_build_s = time.perf_counter() - t0
f = ehx.Space.unique("c1-flat", d, metric=ehx.METRIC_L2SQ, initial_capacity=n)
f.set_batch([b"%d" % i for i in range(n)], X)
out = {"workload": "10000x128 L2 (EHX-GAUSS-1), one query per call, k=10", "oracle_build_s": round(oracle_build_s, 2),
       "gpu_sequential_build_s": round(gpu_build_s, 2), "points": []}
for ef in (10, 200):
    h.set_ef(ef)
    g.set_ef(ef)
    g.knn(Q[:1], k)
    lat, same, hits = [], 0, 0
    for i in range(nq):
        t0 = time.perf_counter()
        ids, dist, cnt = g.knn(Q[i:i + 1], k)
        lat.append(time.perf_counter() - t0)
        o_ids, o_dist = h.search(Q[i], k)
        same += int(np.array_equal(ids[0], o_ids) and dist[0].tobytes() == o_dist.tobytes())
        hits += len(set(ids[0].tolist()) & set(truth[i].tolist()))
    t0 = time.perf_counter()
    for i in range(nq):
        h.search(Q[i], k)
    cpu = (time.perf_counter() - t0) / nq
    out["points"].append({"path": "graph", "ef": ef, "gpu_latency_us_median": round(float(np.median(lat)) * 1e6, 1),
                          "gpu_latency_us_p99": round(float(np.quantile(lat, 0.99)) * 1e6, 1),
                          "cpu_oracle_latency_us_1_thread": round(cpu * 1e6, 1),
                          "queries_identical_to_oracle": "%d of %d" % (same, nq), "recall_at_10": round(hits / (nq * k), 4)})
f.knn(Q[:1], k)
lat = []
for i in range(nq):
    t0 = time.perf_counter()
    ids, dist, cnt = f.knn(Q[i:i + 1], k)
    lat.append(time.perf_counter() - t0)
    assert np.array_equal(ids[0], truth[i])
t0 = time.perf_counter()
pyoracle.exhaustive(X, Q[:100], k, pyoracle.METRIC_L2)
cpu_flat = (time.perf_counter() - t0) / 100
out["points"].append({"path": "flat (exact)", "gpu_latency_us_median": round(float(np.median(lat)) * 1e6, 1),
                      "gpu_latency_us_p99": round(float(np.quantile(lat, 0.99)) * 1e6, 1), "recall_at_10": 1.0,
                      "cpu_oracle_exhaustive_us_per_query": round(cpu_flat * 1e6, 1),
                      "small_request_route": os.environ.get("EHX_SMALL_EXACT_BYTES", "default (512 MiB)"),
                      "engine_counters": {k2: v for k2, v in f.stats().items() if k2.startswith("n_")}})
# the same request shape on the largest shard the small-request route takes (1 M x 128 fp32 = 512 MB)
n2 = 1_000_000
f2 = ehx.Space.unique("c1-flat-1m", d, metric=ehx.METRIC_L2SQ, initial_capacity=n2)
f2.fill_synthetic(ehx.SEED_CORPUS, 0, n2, False)
f2.knn(Q[:1], k)
lat = []
for i in range(200):
    t0 = time.perf_counter()
    f2.knn(Q[i:i + 1], k)
    lat.append(time.perf_counter() - t0)
out["points"].append({"path": "flat (exact), 1M x 128", "gpu_latency_us_median": round(float(np.median(lat)) * 1e6, 1),
                      "gpu_latency_us_p99": round(float(np.quantile(lat, 0.99)) * 1e6, 1)})
print(json.dumps(out))
