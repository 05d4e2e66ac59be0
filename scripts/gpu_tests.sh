#!/bin/bash
# One GPU-box session: the GPU parity tests (all, or the files given), log to gpurun_out/.
# usage: gpurun --timeout 900 -- bash scripts/gpu_tests.sh [pytest args]
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 850 python -m pytest ${@:-tests} -m gpu -q --timeout=300 > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?"
tail -25 gpurun_out/pytest_gpu.log
