#!/bin/bash
# graph build rate of the given library builds ("-" = shipped) + the graph parity tests for the shipped one
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest graph"; timeout 600 python -m pytest tests/test_graph_parity.py -m gpu -x -q --timeout=180 > gpurun_out/pytest_graph.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_graph.log
for v in "$@"; do
  if [ "$v" != "-" ]; then export EHX_LIB=$PWD/embeddinghub_amd/lib/libehx$v.so; else unset EHX_LIB; fi
  echo "== lib='$v'"
  timeout 200 python scripts/bench_graph.py --rows 1000000 --dims 128 --metric l2 --gpu-build --efs 200 --reps 5 > gpurun_out/bab_1m128$v.jsonl 2> gpurun_out/bab$v.err; cut -c1-150 gpurun_out/bab_1m128$v.jsonl
  timeout 300 python scripts/bench_graph.py --rows 300000 --dims 768 --metric cosine --gpu-build --efs 800 --reps 5 > gpurun_out/bab_300k768$v.jsonl 2>> gpurun_out/bab$v.err; cut -c1-150 gpurun_out/bab_300k768$v.jsonl
done
