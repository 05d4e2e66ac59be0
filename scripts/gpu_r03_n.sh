#!/bin/bash
# round 3, session n: same-box A/B — accumulators started from the MFMA's zero operand (current) against zeroed with v_mov (libehx_prev)
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$(pwd)
run() { EHX_LIB=$R/embeddinghub_amd/lib/libehx$1.so timeout 300 python bench.py "${@:2}" --steps 20 --warmup 5 --no-cpu-baseline --no-f32-engine --graph-rows 0 --structured-rows 0 --set-concurrent 0 --single-query 0 --check-queries 0 > gpurun_out/r03_n_tmp.json 2> gpurun_out/r03_n_tmp.err; python - "lib$1" "${@:2}" <<P
import json, sys
j = json.load(open("gpurun_out/r03_n_tmp.json"))
print(" ".join(sys.argv[1:]), "| ms_per_step", j["ms_per_step"], "kernel_ms", j["roofline"]["kernel_ms"], "frac", j["roofline"]["frac"], "| fallback", j["i8_fallback_queries"], "identical", j["exactness"].get("filter_vs_f32_engine_identical"))
P
}
for rep in 1 2; do for lib in _prev ""; do run "$lib" --rows 10000000; done; done
for lib in _prev ""; do run "$lib" --rows 6250000 --dims 128 --metric-kind l2; run "$lib" --rows 4000000 --dims 384; run "$lib" --rows 1250000; done
timeout 900 python -m pytest tests/test_i8_filter.py tests/test_flat_parity.py tests/test_exactness.py -m gpu -q > gpurun_out/r03_n_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r03_n_tests.log
