#!/bin/bash
# Data points for the other BASELINE.json configs (bench.py's default run is the headline, configs[2]'s
# workload on the exact path).  One JSON line each -> gpurun_out/bench_configs.jsonl
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out/bench_configs.jsonl
: > $O
run() { echo "== $*"; timeout 600 python bench.py "$@" --no-cpu-baseline --graph-rows 0 --structured-rows 0 --set-concurrent 0 --single-query 0 2>&1 | tail -1 >> $O; tail -1 $O | cut -c1-200; }
# configs[1]: 1M x 768 cosine, batch 1024, brute force on 1 GPU
run --rows 1000000 --steps 10 --warmup 2
# configs[3]: 50M x 128 L2 over 8 GPUs -> one shard = 6.25M x 128 L2
run --rows 6250000 --dims 128 --metric-kind l2 --steps 10 --warmup 2
# configs[4]: 100M x 1536 fp16 over 8 GPUs -> one shard = 12.5M x 1536, fp16 rows; streamed Set of 200k rows
run --rows 12500000 --dims 1536 --rows-dtype f16 --steps 5 --warmup 1 --set-stream 200000
