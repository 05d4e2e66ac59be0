#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
timeout 300 python scripts/c1_single_query.py 2>gpurun_out/c1.err | tail -1 > gpurun_out/r03_w_c1_single_query.json; tail -2 gpurun_out/c1.err; cat gpurun_out/r03_w_c1_single_query.json
EHX_SMALL_EXACT_BYTES=0 timeout 300 python scripts/c1_single_query.py 2>/dev/null | tail -1 > gpurun_out/r03_w_c1_single_query_route_off.json; cat gpurun_out/r03_w_c1_single_query_route_off.json
