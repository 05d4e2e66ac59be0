#!/bin/bash
# round 3, session s: re-rank with three row chunks in flight: tests, sizes, launch-by-launch trace of a 1 M-row batch
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
R=$(pwd)
timeout 900 python -m pytest tests/test_i8_filter.py tests/test_flat_parity.py tests/test_exactness.py -m gpu -q > gpurun_out/r03_s_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r03_s_tests.log
for rows in 10000000 1250000 1000000; do
  timeout 300 python bench.py --rows $rows --steps 20 --warmup 5 --no-cpu-baseline --no-f32-engine --graph-rows 0 --structured-rows 0 --set-concurrent 0 --single-query 0 --check-queries 0 > gpurun_out/r03_s_rows_$rows.json 2> gpurun_out/r03_s_rows_$rows.err
  python - <<P
import json
j = json.load(open("gpurun_out/r03_s_rows_$rows.json"))
print("rows $rows: ms_per_step", j["ms_per_step"], "q/s", j["value"], "kernel_ms", j["roofline"]["kernel_ms"], "frac", j["roofline"]["frac"], "i8 fallback", j["i8_fallback_queries"], "identical to f32 engine", j["exactness"].get("filter_vs_f32_engine_identical"))
P
done
rm -rf gpurun_out/prof/r03_s_1m
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/prof/r03_s_1m -o trace -- python $R/bench.py --rows 1000000 --steps 6 --warmup 2 --no-cpu-baseline --no-f32-engine --graph-rows 0 --structured-rows 0 --set-concurrent 0 --single-query 0 --check-queries 0 > $R/gpurun_out/prof/r03_s_1m.log 2>&1)
ROCPD_SEQ=14 python scripts/rocpd_summary.py gpurun_out/prof/r03_s_1m > gpurun_out/r03_s_trace_1m_rows_summary.txt 2>&1; grep -E "^#|Rerank|select256" gpurun_out/r03_s_trace_1m_rows_summary.txt | tail -18 | cut -c1-130
