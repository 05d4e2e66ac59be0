#!/bin/bash
# round 3, session i: int8 rows padded to stage pairs (d = 128 / 384 / 64 / 640 on the int8 engine), round shares
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_i8_filter.py tests/test_flat_parity.py tests/test_exactness.py tests/test_graph_scale.py tests/test_graph_parity.py -m gpu -q -x > gpurun_out/r03_i_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r03_i_tests.log
run() { timeout 300 python bench.py "$@" --steps 20 --warmup 5 --no-cpu-baseline --no-f32-engine --graph-rows 0 --structured-rows 0 --set-concurrent 0 --single-query 0 --check-queries 0 > gpurun_out/r03_i_tmp.json 2> gpurun_out/r03_i_tmp.err; python - "$@" <<P
import json, sys
j = json.load(open("gpurun_out/r03_i_tmp.json"))
print(" ".join(sys.argv[1:]), "| ms_per_step", j["ms_per_step"], "q/s", j["value"], "|", j["roofline"]["kernel"][:20], "frac", j["roofline"]["frac"], "of", j["roofline"]["peak"], "| fallback", j["i8_fallback_queries"], j["filter_fallback_queries"], "identical", j["exactness"].get("filter_vs_f32_engine_identical"))
open("gpurun_out/r03_i_configs.jsonl", "a").write(json.dumps(j) + "\n")
P
}
rm -f gpurun_out/r03_i_configs.jsonl
run --rows 10000000
run --rows 6250000 --dims 128 --metric-kind l2
run --rows 6250000 --dims 128 --metric-kind l2 --scan f16
run --rows 4000000 --dims 384
run --rows 1000000 --dims 128 --metric-kind cosine
EHX_BUILD_TRACE=1 timeout 300 python scripts/bench_graph.py --rows 2000000 --dims 768 --metric cosine --gpu-build --efs 100,400 > gpurun_out/r03_i_graph_2m768.jsonl 2> gpurun_out/r03_i_graph_2m768.err; grep "ehx build" gpurun_out/r03_i_graph_2m768.err | tail -1; python scripts/jl.py gpurun_out/r03_i_graph_2m768.jsonl | cut -c1-140
