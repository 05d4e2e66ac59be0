#!/bin/bash
# Phase timers of the graph search (ablation build libehx_gprof.so = -DEHX_GRAPH_PROFILE).
mkdir -p gpurun_out
export TMPDIR=/tmp
export EHX_LIB=$PWD/embeddinghub_amd/lib/libehx_gprof.so
timeout 200 python scripts/bench_graph.py --rows 1000000 --dims 128 --metric l2 --gpu-build --efs 50,200,800 > gpurun_out/gprof_1m128.jsonl 2> gpurun_out/gprof_1m128.err; echo "rc=$?"; python scripts/jl.py gpurun_out/gprof_1m128.jsonl | cut -c1-40,150-
timeout 200 python scripts/bench_graph.py --rows 300000 --dims 768 --metric cosine --gpu-build --efs 100,800 > gpurun_out/gprof_300k768.jsonl 2> gpurun_out/gprof_300k768.err; echo "rc=$?"; python scripts/jl.py gpurun_out/gprof_300k768.jsonl | cut -c1-40,150-
