#!/bin/bash
# Graph-path session: graph parity tests, then the graph bench on the two r01_d datasets.
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest graph"; timeout 600 python -m pytest tests/test_graph_parity.py tests/test_abi.py -m gpu -x -q --timeout=180 > gpurun_out/pytest_graph.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/pytest_graph.log
echo "== graph bench 1M x 128"; timeout 200 python scripts/bench_graph.py --rows 1000000 --dims 128 --metric l2 --gpu-build --efs 50,200,800 > gpurun_out/graph_1m128.jsonl 2> gpurun_out/graph_1m128.err; echo "rc=$?"; python scripts/jl.py gpurun_out/graph_1m128.jsonl
echo "== graph bench 300k x 768"; timeout 200 python scripts/bench_graph.py --rows 300000 --dims 768 --metric cosine --gpu-build --efs 100,800 > gpurun_out/graph_300k768.jsonl 2> gpurun_out/graph_300k768.err; echo "rc=$?"; python scripts/jl.py gpurun_out/graph_300k768.jsonl
tail -3 gpurun_out/graph_1m128.err gpurun_out/graph_300k768.err
