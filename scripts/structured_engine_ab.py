#!/usr/bin/env python3
"""Which exact engine serves structured rows fastest?  EHX-MANIFOLD-1 rows generated on the device, device-resident query
batches, the three engines of one flat space in turn (int8 filter / fp16 filter / fp32 scan: identical answers), ms per
batch from the engine's own HIP events.  One JSON line per engine.
    python scripts/structured_engine_ab.py --rows 6250000 --dims 128 --metric l2 --latent 16"""
import argparse
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=6_250_000)
    ap.add_argument("--dims", type=int, default=128)
    ap.add_argument("--metric", default="l2")
    ap.add_argument("--latent", type=int, default=16, help="0 = EHX-GAUSS-1 rows")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--engines", default="auto,f16,f32")
    a = ap.parse_args()
    import torch
    import embeddinghub_amd as ehx
    from embeddinghub_amd import _lib
    L = _lib.load()
    _lib.check(L.ehx_init((C.c_int * 1)(0), 1))
    metric = {"cosine": ehx.METRIC_COSINE, "l2": ehx.METRIC_L2SQ, "ip": ehx.METRIC_IP}[a.metric]
    norm = a.metric == "cosine"
    sp = ehx.Space("seab", a.dims, metric=metric, initial_capacity=a.rows)
    if a.latent:
        sp.fill_manifold(ehx.SEED_CORPUS, 0, a.rows, a.latent, norm)
    else:
        sp.fill_synthetic(ehx.SEED_CORPUS, 0, a.rows, norm)
    B, k, nb = 1024, 10, 4
    st = torch.cuda.current_stream().cuda_stream
    q = torch.empty((nb, B, a.dims), dtype=torch.float32, device="cuda")
    for i in range(nb):
        if a.latent:
            _lib.check(L.ehx_gen_manifold_rows_device(C.c_void_p(st), ehx.SEED_QUERY, i * B, B, a.dims, a.latent, int(norm),
                                                      C.c_void_p(q[i].data_ptr())))
        else:
            _lib.check(L.ehx_gen_rows_device(C.c_void_p(st), ehx.SEED_QUERY, i * B, B, a.dims, int(norm), C.c_void_p(q[i].data_ptr())))
    ids = torch.empty((B, k), dtype=torch.int64, device="cuda")
    dst = torch.empty((B, k), dtype=torch.float32, device="cuda")
    cnt = torch.empty((B,), dtype=torch.int32, device="cuda")
    ref = None
    for eng in a.engines.split(","):
        sp.set_scan({"auto": ehx.SCAN_AUTO, "f16": ehx.SCAN_F16, "f32": ehx.SCAN_F32}[eng])
        for i in range(4):
            sp.knn_device(q[i % nb], k, ids, dst, cnt, stream=st)
        torch.cuda.synchronize()
        sp.stats_reset()
        t0 = time.perf_counter()
        for i in range(a.steps):
            sp.knn_device(q[i % nb], k, ids, dst, cnt, stream=st)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        s = sp.stats()
        chk = (int(ids.sum().item()), float(dst.double().sum().item()))
        same = ref is None or chk == ref
        ref = ref or chk
        print(json.dumps({"rows": a.rows, "dims": a.dims, "metric": a.metric, "latent": a.latent, "engine_selected": eng,
                          "engine_that_ran": sp.scan_engine(), "ms_per_batch": round(el / a.steps * 1e3, 3),
                          "scan_ms_mean": round(s["scan_ms_mean"], 3), "i8_fallback": int(s.get("n_i8_fallback", 0)),
                          "filter_fallback": int(s["n_filter_fallback"]), "exhaustive": int(s["n_exhaustive"]),
                          "same_answer_as_first_engine": same}), flush=True)


if __name__ == "__main__":
    main()
