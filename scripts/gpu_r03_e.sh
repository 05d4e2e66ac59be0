#!/bin/bash
# round 3, session e: what an integer phase 1 would buy; the last pass's threshold rank (EHX_I8_SAFETY); graph scale
# tests with the default round shares
mkdir -p gpurun_out
export TMPDIR=/tmp
./scripts/ablate_i8.sh 10000000 "" _abl2 _abl16 > gpurun_out/r03_e_ablate_i8.txt 2>&1; cat gpurun_out/r03_e_ablate_i8.txt
for sf in 4 2 1.5 1; do
  EHX_I8_SAFETY=$sf timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-engine --graph-rows 0 --structured-rows 0 --set-concurrent 0 --single-query 0 --check-queries 0 > gpurun_out/r03_e_safety_$sf.json 2> gpurun_out/r03_e_safety_$sf.err
  python - <<P
import json
j = json.load(open("gpurun_out/r03_e_safety_$sf.json"))
print("safety $sf: ms_per_step", j["ms_per_step"], "kernel_ms", j["roofline"]["kernel_ms"], "frac", j["roofline"]["frac"], "i8 fallback", j["i8_fallback_queries"], "uncertified", j["n_uncertified"], "filter_vs_f32", j["exactness"].get("filter_vs_f32_engine_identical"))
P
done
EHX_SCALE_REPORT=$PWD/gpurun_out/r03_e_scale_report.jsonl timeout 900 python -m pytest tests/test_graph_scale.py -m gpu -q > gpurun_out/r03_e_scale.log 2>&1; echo "scale rc=$?"; tail -4 gpurun_out/r03_e_scale.log; cat gpurun_out/r03_e_scale_report.jsonl
