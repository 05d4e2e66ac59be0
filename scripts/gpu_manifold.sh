#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest"; timeout 600 python -m pytest tests/test_graph_parity.py -m gpu -x -q --timeout=180 > gpurun_out/pytest_graph.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/pytest_graph.log
timeout 300 python scripts/bench_graph.py --rows 1000000 --dims 128 --metric l2 --data manifold:16 --efs 10,20,40,80 --reps 5 > gpurun_out/graph_manifold16_1m128.jsonl 2> gpurun_out/mf1.err; echo "rc=$?"; python scripts/jl.py gpurun_out/graph_manifold16_1m128.jsonl | cut -c1-330; cut -c1-200 gpurun_out/graph_manifold16_1m128.jsonl | head -1
timeout 400 python scripts/bench_graph.py --rows 1000000 --dims 768 --metric cosine --data manifold:32 --efs 10,20,40,80 --reps 5 > gpurun_out/graph_manifold32_1m768.jsonl 2> gpurun_out/mf2.err; echo "rc=$?"; python scripts/jl.py gpurun_out/graph_manifold32_1m768.jsonl | cut -c1-330; cut -c1-200 gpurun_out/graph_manifold32_1m768.jsonl | head -1
tail -n 4 gpurun_out/mf1.err gpurun_out/mf2.err
