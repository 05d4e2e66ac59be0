#!/bin/bash
# A/B of graph-search ablation builds: pass the lib suffixes as arguments ("" = shipped build)
mkdir -p gpurun_out
export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" != "-" ]; then export EHX_LIB=$PWD/embeddinghub_amd/lib/libehx$v.so; else unset EHX_LIB; v=""; fi
  echo "== lib='$v'"
  timeout 200 python scripts/bench_graph.py --rows 1000000 --dims 128 --metric l2 --gpu-build --efs 200,800 --reps 10 > gpurun_out/ab_1m128$v.jsonl 2> gpurun_out/ab$v.err; python scripts/jl.py gpurun_out/ab_1m128$v.jsonl | cut -c1-150
  timeout 200 python scripts/bench_graph.py --rows 300000 --dims 768 --metric cosine --gpu-build --efs 800 --reps 10 > gpurun_out/ab_300k768$v.jsonl 2>> gpurun_out/ab$v.err; python scripts/jl.py gpurun_out/ab_300k768$v.jsonl | cut -c1-150
done
