#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_i8_filter.py tests/test_graph_parity.py -m gpu -q --timeout=300 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest_gpu.log
S="--graph-rows 0 --structured-rows 0 --no-cpu-baseline --no-f32-engine --check-queries 0 --steps 4 --warmup 1"
EHX_I8_DEBUG=1 EHX_I8_SAFETY=4 timeout 300 python bench.py $S > gpurun_out/bench_dbg.log 2> gpurun_out/bench_dbg.err; grep "i8 debug" gpurun_out/bench_dbg.err | head -24
