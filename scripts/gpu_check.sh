#!/bin/bash
# One GPU-box session: smoke, GPU parity tests, a short bench.  Everything is bounded by `timeout`
# so a hung kernel cannot hold the box; logs land in gpurun_out/.
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== smoke"; timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/smoke.log
echo "== pytest -m gpu"; timeout 900 python -m pytest tests -m gpu -q --timeout=180 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/pytest_gpu.log
echo "== bench 1M"; timeout 300 python bench.py --rows 1000000 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_1m.log 2>&1; echo "bench1m rc=$?"; tail -1 gpurun_out/bench_1m.log | cut -c1-400
