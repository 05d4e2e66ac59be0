#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest graph"; timeout 600 python -m pytest tests/test_graph_parity.py -m gpu -x -q --timeout=180 > gpurun_out/pytest_graph.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/pytest_graph.log
for v in "" _gprof; do
  if [ -n "$v" ]; then export EHX_LIB=$PWD/embeddinghub_amd/lib/libehx$v.so; fi
  echo "== graph bench lib='$v'"
  timeout 200 python scripts/bench_graph.py --rows 1000000 --dims 128 --metric l2 --gpu-build --efs 50,200,800 > gpurun_out/graph_1m128$v.jsonl 2> gpurun_out/graph_1m128$v.err; echo "rc=$?"; python scripts/jl.py gpurun_out/graph_1m128$v.jsonl
  timeout 200 python scripts/bench_graph.py --rows 300000 --dims 768 --metric cosine --gpu-build --efs 100,800 > gpurun_out/graph_300k768$v.jsonl 2> gpurun_out/graph_300k768$v.err; echo "rc=$?"; python scripts/jl.py gpurun_out/graph_300k768$v.jsonl
done
