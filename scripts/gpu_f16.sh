#!/bin/bash
# fp16-row storage checks on the GPU box: parity tests, then the bench workload with fp16 rows.
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== fp16 parity"; timeout 600 python -m pytest tests/test_flat_parity.py -m gpu -q --timeout=120 -k "fp16 or random_parity or full_size" > gpurun_out/pytest_f16.log 2>&1; echo "rc=$?"; tail -8 gpurun_out/pytest_f16.log
echo "== bench 2M f32"; timeout 300 python bench.py --rows 2000000 --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/bench_2m_f32.log 2>&1; tail -1 gpurun_out/bench_2m_f32.log
echo "== bench 2M f16"; timeout 300 python bench.py --rows 2000000 --steps 5 --warmup 1 --no-cpu-baseline --rows-dtype f16 > gpurun_out/bench_2m_f16.log 2>&1; tail -1 gpurun_out/bench_2m_f16.log
echo "== bench 10M f16"; timeout 400 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --rows-dtype f16 > gpurun_out/bench_10m_f16.log 2>&1; tail -1 gpurun_out/bench_10m_f16.log
