#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
bash scripts/gpu_graph_profile.sh
unset EHX_LIB
timeout 200 python scripts/bench_graph.py --rows 1000000 --dims 128 --metric l2 --gpu-build --efs 50,200,800 > gpurun_out/graph_1m128.jsonl 2> gpurun_out/g1.err; python scripts/jl.py gpurun_out/graph_1m128.jsonl | cut -c1-150
timeout 200 python scripts/bench_graph.py --rows 300000 --dims 768 --metric cosine --gpu-build --efs 100,800 > gpurun_out/graph_300k768.jsonl 2> gpurun_out/g2.err; python scripts/jl.py gpurun_out/graph_300k768.jsonl | cut -c1-150
timeout 300 python scripts/bench_graph.py --rows 2000000 --dims 768 --metric cosine --gpu-build --efs 100,400,1600 --reps 3 > gpurun_out/graph_2m768.jsonl 2> gpurun_out/g3.err; python scripts/jl.py gpurun_out/graph_2m768.jsonl | cut -c1-150
timeout 500 python scripts/bench_graph.py --rows 10000000 --dims 768 --metric cosine --gpu-build --efs 100,400,1600 --reps 3 > gpurun_out/graph_10m768.jsonl 2> gpurun_out/g4.err; python scripts/jl.py gpurun_out/graph_10m768.jsonl | cut -c1-150
timeout 300 python scripts/bench_graph.py --rows 1000000 --dims 768 --metric cosine --data manifold:32 --efs 10,20,40,80 --reps 5 > gpurun_out/graph_manifold32_1m768.jsonl 2> gpurun_out/g5.err; python scripts/jl.py gpurun_out/graph_manifold32_1m768.jsonl | cut -c1-150
timeout 300 python scripts/bench_graph.py --rows 1000000 --dims 128 --metric l2 --data manifold:16 --efs 10,20,40,80 --reps 5 > gpurun_out/graph_manifold16_1m128.jsonl 2> gpurun_out/g6.err; python scripts/jl.py gpurun_out/graph_manifold16_1m128.jsonl | cut -c1-150
