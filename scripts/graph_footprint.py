#!/usr/bin/env python3
"""HBM held by a GPU-built graph space (free-memory counter before / after), then one search batch against the exact flat
engine over the same rows (the bulk build releases its scratch: the search re-allocates its visited bitmaps)."""
import argparse
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=4_000_000)
    ap.add_argument("--dims", type=int, default=768)
    ap.add_argument("--metric", default="cosine")
    a = ap.parse_args()
    import torch
    import embeddinghub_amd as ehx
    from embeddinghub_amd import _lib
    L = _lib.load()
    _lib.check(L.ehx_init((C.c_int * 1)(0), 1))
    metric = {"cosine": ehx.METRIC_COSINE, "l2": ehx.METRIC_L2SQ, "ip": ehx.METRIC_IP}[a.metric]
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    g = ehx.Space("fp", a.dims, metric=metric, mode=ehx.MODE_GRAPH, initial_capacity=a.rows)
    t0 = time.perf_counter()
    g.fill_synthetic(ehx.SEED_CORPUS, 0, a.rows, True)
    build_s = time.perf_counter() - t0
    torch.cuda.synchronize()
    built = free0 - torch.cuda.mem_get_info()[0]
    B, k = 1024, 10
    st = torch.cuda.current_stream().cuda_stream
    q = torch.empty((B, a.dims), dtype=torch.float32, device="cuda")
    _lib.check(L.ehx_gen_rows_device(C.c_void_p(st), ehx.SEED_QUERY, 0, B, a.dims, 1, C.c_void_p(q.data_ptr())))
    ids = torch.empty((B, k), dtype=torch.int64, device="cuda")
    dst = torch.empty((B, k), dtype=torch.float32, device="cuda")
    cnt = torch.empty((B,), dtype=torch.int32, device="cuda")
    g.set_ef(400)
    g.knn_device(q, k, ids, dst, cnt, stream=st)
    torch.cuda.synchronize()
    searching = free0 - torch.cuda.mem_get_info()[0]
    got = ids.cpu().numpy().copy()
    f = ehx.Space("fp-flat", a.dims, metric=metric, initial_capacity=a.rows)
    f.fill_synthetic(ehx.SEED_CORPUS, 0, a.rows, True)
    f.knn_device(q, k, ids, dst, cnt, stream=st)
    torch.cuda.synchronize()
    truth = ids.cpu().numpy()
    rec = sum(len(set(got[j]) & set(truth[j])) for j in range(B)) / (B * k)
    print(json.dumps({"rows": a.rows, "dims": a.dims, "metric": a.metric, "build_s": round(build_s, 1),
                      "rows_fp32_GB": round(a.rows * a.dims * 4 / 1e9, 2), "hbm_after_build_GB": round(built / 1e9, 2),
                      "hbm_after_first_1024_query_search_GB": round(searching / 1e9, 2),
                      "recall_at_10_ef400_vs_exact": round(rec, 4),
                      "two_copies": os.environ.get("EHX_GRAPH_TWO_COPIES", "0")}), flush=True)


if __name__ == "__main__":
    main()
