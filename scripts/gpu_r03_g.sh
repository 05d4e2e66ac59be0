#!/bin/bash
# round 3, session g: adaptive list width, first-pass key target, pinned verdict; configs[3] / configs[4] shards
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_i8_filter.py tests/test_flat_parity.py tests/test_exactness.py -m gpu -q -x > gpurun_out/r03_g_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r03_g_tests.log
for rows in 10000000 1250000 1000000; do
  timeout 300 python bench.py --rows $rows --steps 20 --warmup 5 --no-cpu-baseline --no-f32-engine --graph-rows 0 --structured-rows 0 --set-concurrent 0 --single-query 0 --check-queries 0 > gpurun_out/r03_g_rows_$rows.json 2> gpurun_out/r03_g_rows_$rows.err
  python - <<P
import json
j = json.load(open("gpurun_out/r03_g_rows_$rows.json"))
print("rows $rows: ms_per_step", j["ms_per_step"], "q/s", j["value"], "kernel_ms", j["roofline"]["kernel_ms"], "frac", j["roofline"]["frac"], "i8 fallback", j["i8_fallback_queries"], "identical to f32 engine", j["exactness"].get("filter_vs_f32_engine_identical"))
P
done
EHX_I8_TRACE=1 timeout 600 python bench.py --rows 12500000 --dims 1536 --rows-dtype f16 --steps 10 --warmup 4 --no-cpu-baseline --no-f32-engine --graph-rows 0 --structured-rows 0 --set-concurrent 0 --single-query 0 --check-queries 0 > gpurun_out/r03_g_c5_shard.json 2> gpurun_out/r03_g_c5_shard.err; grep "ehx i8" gpurun_out/r03_g_c5_shard.err
python - <<P
import json
j = json.load(open("gpurun_out/r03_g_c5_shard.json"))
print("c5 shard: ms_per_step", j["ms_per_step"], "q/s", j["value"], "kernel_ms", j["roofline"]["kernel_ms"], "frac", j["roofline"]["frac"], "i8 fallback", j["i8_fallback_queries"], "of", j["i8_queries"], "identical", j["exactness"].get("filter_vs_f32_engine_identical"))
P
timeout 600 python scripts/bench_graph.py --rows 6250000 --dims 128 --metric l2 --gpu-build --efs 50,200,800 > gpurun_out/r03_g_graph_6250k128.jsonl 2> gpurun_out/r03_g_graph_6250k128.err; echo "graph rc=$?"; python scripts/jl.py gpurun_out/r03_g_graph_6250k128.jsonl | cut -c1-220
