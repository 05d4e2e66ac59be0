#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
timeout 300 python scripts/c1_single_query.py 2>/dev/null | tail -1 > gpurun_out/r03_x_c1_single_query.json; cat gpurun_out/r03_x_c1_single_query.json
timeout 600 python scripts/bench_graph.py --rows 6250000 --dims 128 --metric l2 --gpu-build --efs 50,200 --batches 1024,2048,4096,8192 --reps 5 2>gpurun_out/x_graph.err | grep '^{' > gpurun_out/r03_x_graph_6250k128_batches.jsonl; tail -3 gpurun_out/x_graph.err; cut -c1-400 gpurun_out/r03_x_graph_6250k128_batches.jsonl
