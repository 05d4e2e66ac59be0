#!/bin/bash
# round 3, session j: same-box A/B of the int8 kernel with run-time ring slots (current) against compile-time slots (libehx_prev)
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$(pwd)
for rep in 1 2; do for lib in _prev ""; do
  EHX_LIB=$R/embeddinghub_amd/lib/libehx$lib.so timeout 300 python bench.py --rows 10000000 --steps 20 --warmup 5 --no-cpu-baseline --no-f32-engine --graph-rows 0 --structured-rows 0 --set-concurrent 0 --single-query 0 --check-queries 0 > gpurun_out/r03_j_tmp.json 2> gpurun_out/r03_j_tmp.err
  python - "lib$lib" <<P
import json, sys
j = json.load(open("gpurun_out/r03_j_tmp.json"))
print(sys.argv[1], "| ms_per_step", j["ms_per_step"], "kernel_ms", j["roofline"]["kernel_ms"], "frac", j["roofline"]["frac"], "| fallback", j["i8_fallback_queries"], "identical", j["exactness"].get("filter_vs_f32_engine_identical"))
P
done; done
timeout 900 python -m pytest tests/test_i8_filter.py tests/test_flat_parity.py tests/test_exactness.py -m gpu -q > gpurun_out/r03_j_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r03_j_tests.log
