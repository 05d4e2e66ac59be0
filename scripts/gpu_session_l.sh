#!/bin/bash
# graph kernel: parity tests on the shipped build, then A/B of the ablation builds given as arguments, then phase timers
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_graph_parity.py -m gpu -x -q --timeout=300 > gpurun_out/pytest_graph.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_graph.log
bash scripts/gpu_graph_ab.sh "$@"
bash scripts/gpu_graph_profile.sh
