#!/bin/bash
# The graph path on rows with structure, by intrinsic dimension, with INDEPENDENT queries (DESIGN.md §e erratum): for
# R = 8 / 16 / 24 / 32 a 1 M x 768 cosine index built on the GPU, recall-vs-ef against the exact flat engine of the same
# rows, kernel rates, HBM roofline.  One JSON line per (R, ef) -> gpurun_out/structured_sweep.jsonl   (~3 GPU-minutes)
# usage: gpurun --timeout 900 -- bash scripts/gpu_structured_sweep.sh
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out/structured_sweep.jsonl
: > $O
for R in 8 16 24 32; do
  timeout 400 python scripts/bench_graph.py --rows ${ROWS:-1000000} --dims 768 --metric cosine --gpu-build \
      --data manifold:$R --efs 10,20,40,100,200,400 --reps 5 2>gpurun_out/structured_sweep_$R.err | grep '^{' >> $O
done
python - <<'PY'
import json
for l in open("gpurun_out/structured_sweep.jsonl"):
    j = json.loads(l)
    print(j["workload"][:70], "| recall %.3f | %.0f q/s kernel | %.0f q/s exact flat | %.0f rows/query | %.2f of 8 TB/s" % (
        j["recall_at_k"], j["qps_kernel"], j.get("flat_exact_qps_host_pointers") or 0, j["n_dist_per_query"], j["roofline"]["frac"]))
PY
