#!/usr/bin/env python3
"""ehx_knn from host pointers with 1 .. N caller threads against device-resident batches (ehx_knn_device): queries/s and
ms per batch of each, one JSON line.  (bench.py's HostCallers on an arbitrary shape; EHX_HOST_PIPELINE=0: the callers'
device pipelines strictly one after the other, as before round 4.)"""
import argparse
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--dims", type=int, default=768)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--callers", default="1,2,3")
    a = ap.parse_args()
    import numpy as np
    import torch
    import embeddinghub_amd as ehx
    from embeddinghub_amd import _lib
    L = _lib.load()
    _lib.check(L.ehx_init((C.c_int * 1)(0), 1))
    sp = ehx.Space("hc", a.dims, metric=ehx.METRIC_COSINE, initial_capacity=a.rows)
    sp.fill_synthetic(ehx.SEED_CORPUS, 0, a.rows, True)
    B, k, nb = 1024, 10, 8
    st = torch.cuda.current_stream().cuda_stream
    q = torch.empty((nb, B, a.dims), dtype=torch.float32, device="cuda")
    for i in range(nb):
        _lib.check(L.ehx_gen_rows_device(C.c_void_p(st), ehx.SEED_QUERY, i * B, B, a.dims, 1, C.c_void_p(q[i].data_ptr())))
    torch.cuda.synchronize()
    hq = [np.ascontiguousarray(q[i].cpu().numpy()) for i in range(nb)]
    ids = torch.empty((B, k), dtype=torch.int64, device="cuda")
    dst = torch.empty((B, k), dtype=torch.float32, device="cuda")
    cnt = torch.empty((B,), dtype=torch.int32, device="cuda")
    out = {"rows": a.rows, "dims": a.dims, "steps": a.steps, "pipeline": os.environ.get("EHX_HOST_PIPELINE", "1")}
    for i in range(4):
        sp.knn_device(q[i % nb], k, ids, dst, cnt, stream=st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        sp.knn_device(q[i % nb], k, ids, dst, cnt, stream=st)
    torch.cuda.synchronize()
    out["device_resident_ms"] = round((time.perf_counter() - t0) / a.steps * 1e3, 4)
    for c in [int(x) for x in a.callers.split(",")]:
        hc = bench.HostCallers(sp, hq, k, c)
        hc.run([i % nb for i in range(2 * c)])
        best = None
        for _ in range(3):
            el = hc.run([i % nb for i in range(a.steps)])
            best = el if best is None else min(best, el)
        out["host_%d_callers_ms" % c] = round(best / a.steps * 1e3, 4)
    st_ = sp.stats()
    out["i8_fallback"] = int(st_.get("n_i8_fallback", 0))
    print(json.dumps(out), flush=True)
    sp.drop()


if __name__ == "__main__":
    main()
