#!/bin/bash
# round 3, session c: tile-shape microbenchmark, the whole GPU suite, the default bench as the driver runs it
mkdir -p gpurun_out
export TMPDIR=/tmp
./scripts/ubench/i8_tile_shapes.bin > gpurun_out/r03_c_i8_tile_shapes.txt 2>&1; cat gpurun_out/r03_c_i8_tile_shapes.txt
EHX_SCALE_REPORT=$PWD/gpurun_out/r03_c_scale_report.jsonl timeout 1200 python -m pytest tests -m gpu -q -x > gpurun_out/r03_c_tests.log 2>&1; echo "tests rc=$?"; tail -8 gpurun_out/r03_c_tests.log
EHX_BUILD_TRACE=1 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03_c_bench.json 2> gpurun_out/r03_c_bench.err; echo "bench rc=$?"; grep -v "ehx build" gpurun_out/r03_c_bench.err | tail -12; grep "ehx build" gpurun_out/r03_c_bench.err | tail -2; cut -c1-600 gpurun_out/r03_c_bench.json
