#!/usr/bin/env python3
"""Same-box A/B of the exact flat path: one shape, device-resident query batches, ms per batch and the engine's own
HIP-event scan time.  Run it once per library (EHX_LIB=<path to a libehx.so>) in the same gpurun session:
    EHX_LIB=embeddinghub_amd/lib/libehx_r03.so python scripts/ab_flat.py --rows 10000000 --dims 768
    python scripts/ab_flat.py --rows 10000000 --dims 768
Prints one JSON line per shape.  (Not a parity check: tests/ and bench.py do that.)"""
import argparse
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--dims", type=int, default=768)
    ap.add_argument("--metric", default="cosine")
    ap.add_argument("--f16", action="store_true")
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--label", default="")
    a = ap.parse_args()
    import torch
    import embeddinghub_amd as ehx
    from embeddinghub_amd import _lib
    L = _lib.load()
    _lib.check(L.ehx_init((C.c_int * 1)(0), 1))
    metric = {"cosine": ehx.METRIC_COSINE, "l2": ehx.METRIC_L2SQ, "ip": ehx.METRIC_IP}[a.metric]
    sp = ehx.Space("ab", a.dims, metric=metric, initial_capacity=a.rows, dtype=ehx.DTYPE_F16 if a.f16 else ehx.DTYPE_F32)
    sp.fill_synthetic(ehx.SEED_CORPUS, 0, a.rows, True)
    torch.cuda.synchronize()
    B, k, nb = a.batch, 10, 8
    st = torch.cuda.current_stream().cuda_stream
    q = torch.empty((nb, B, a.dims), dtype=torch.float32, device="cuda")
    for i in range(nb):
        _lib.check(L.ehx_gen_rows_device(C.c_void_p(st), ehx.SEED_QUERY, i * B, B, a.dims, 1, C.c_void_p(q[i].data_ptr())))
    ids = torch.empty((B, k), dtype=torch.int64, device="cuda")
    dst = torch.empty((B, k), dtype=torch.float32, device="cuda")
    cnt = torch.empty((B,), dtype=torch.int32, device="cuda")
    for i in range(a.warmup):
        sp.knn_device(q[i % nb], k, ids, dst, cnt, stream=st)
    torch.cuda.synchronize()
    sp.stats_reset()
    t0 = time.perf_counter()
    for i in range(a.steps):
        sp.knn_device(q[(a.warmup + i) % nb], k, ids, dst, cnt, stream=st)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    s = sp.stats()
    chk = int(ids.sum().item())
    print(json.dumps({"label": a.label, "lib": os.environ.get("EHX_LIB", "default"), "rows": a.rows, "dims": a.dims,
                      "metric": a.metric, "f16": a.f16, "ms_per_step": round(el / a.steps * 1e3, 4),
                      "kernel_ms": round(s["scan_ms_mean"], 4), "engine": sp.scan_engine(),
                      "i8_fallback": int(s.get("n_i8_fallback", 0)), "filter_fallback": int(s["n_filter_fallback"]),
                      "exhaustive": int(s["n_exhaustive"]), "ids_checksum_last_batch": chk}), flush=True)
    sp.drop()


if __name__ == "__main__":
    main()
