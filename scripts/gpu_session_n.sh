#!/bin/bash
# visit log vs per-batch memset of the visited bitmaps (the memset now sits inside the timed region), two index sizes
mkdir -p gpurun_out
export TMPDIR=/tmp
for v in 1 0; do
  export EHX_GRAPH_VISLOG=$v
  echo "== vislog=$v"
  timeout 300 python scripts/bench_graph.py --rows 2000000 --dims 768 --metric cosine --gpu-build --efs 100,400 --reps 5 > gpurun_out/vl${v}_2m768.jsonl 2> gpurun_out/vl.err; python scripts/jl.py gpurun_out/vl${v}_2m768.jsonl | cut -c1-150
  timeout 300 python scripts/bench_graph.py --rows 6250000 --dims 128 --metric l2 --gpu-build --efs 50,200,800 --reps 5 > gpurun_out/vl${v}_6250k128.jsonl 2>> gpurun_out/vl.err; python scripts/jl.py gpurun_out/vl${v}_6250k128.jsonl | cut -c1-150
done
