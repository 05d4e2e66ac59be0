#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest graph"; timeout 600 python -m pytest tests/test_graph_parity.py -m gpu -x -q --timeout=180 > gpurun_out/pytest_graph.log 2>&1; echo "pytest rc=$?"; tail -12 gpurun_out/pytest_graph.log | cut -c1-200
bash scripts/gpu_graph_ab.sh -
bash scripts/gpu_graph_profile.sh
