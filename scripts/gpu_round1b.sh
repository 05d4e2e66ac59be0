#!/bin/bash
# One GPU-box session after the graph-search prefetch change: GPU parity tests, graph bench on the two
# r01_d datasets, default bench.  Everything bounded by `timeout`; logs land in gpurun_out/.
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest -m gpu"; timeout 700 python -m pytest tests -m gpu -x -q --timeout=180 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/pytest_gpu.log
echo "== graph bench 1M x 128"; timeout 200 python scripts/bench_graph.py --rows 1000000 --dims 128 --metric l2 --gpu-build --efs 50,200,800 > gpurun_out/graph_1m128.jsonl 2> gpurun_out/graph_1m128.err; echo "rc=$?"; cut -c1-420 gpurun_out/graph_1m128.jsonl
echo "== graph bench 300k x 768"; timeout 200 python scripts/bench_graph.py --rows 300000 --dims 768 --metric cosine --gpu-build --efs 100,800 > gpurun_out/graph_300k768.jsonl 2> gpurun_out/graph_300k768.err; echo "rc=$?"; cut -c1-420 gpurun_out/graph_300k768.jsonl
echo "== smoke"; timeout 200 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/smoke.log
echo "== bench default"; timeout 400 python bench.py > gpurun_out/bench_default.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/bench_default.log | cut -c1-700
