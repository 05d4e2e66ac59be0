#!/bin/bash
# filter-scan checks on the GPU box: parity tests, then the bench workload with each scan engine.
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== flat parity"; timeout 600 python -m pytest tests/test_flat_parity.py -m gpu -q -x --timeout=120 > gpurun_out/pytest_flat.log 2>&1; echo "rc=$?"; tail -8 gpurun_out/pytest_flat.log
echo "== bench 2M filter"; timeout 300 python bench.py --rows 2000000 --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/bench_2m_f16.log 2>&1; tail -1 gpurun_out/bench_2m_f16.log
echo "== bench 10M filter"; timeout 400 python bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/bench_10m_f16.log 2>&1; tail -1 gpurun_out/bench_10m_f16.log
echo "== bench 10M fp32 scan"; EHX_SCAN=f32 timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_10m_f32.log 2>&1; tail -1 gpurun_out/bench_10m_f32.log
