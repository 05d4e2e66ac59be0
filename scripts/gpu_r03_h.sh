#!/bin/bash
# round 3, session h: rows per build round (4096 / 8192 / 16384) at 2 M x 768 and 2 M x 128: rate and recall
mkdir -p gpurun_out
export TMPDIR=/tmp
for bb in 4096 8192 16384; do
  EHX_BUILD_TRACE=1 timeout 300 python scripts/bench_graph.py --rows 2000000 --dims 768 --metric cosine --gpu-build --build-batch $bb --efs 100,400 > gpurun_out/r03_h_2m768_bb$bb.jsonl 2> gpurun_out/r03_h_2m768_bb$bb.err
  echo "== 2M x 768, rounds of $bb: $(grep 'ehx build' gpurun_out/r03_h_2m768_bb$bb.err | tail -1)"; python scripts/jl.py gpurun_out/r03_h_2m768_bb$bb.jsonl | cut -c1-140
done
for bb in 4096 16384; do
  EHX_BUILD_TRACE=1 timeout 300 python scripts/bench_graph.py --rows 2000000 --dims 128 --metric l2 --gpu-build --build-batch $bb --efs 50,200 > gpurun_out/r03_h_2m128_bb$bb.jsonl 2> gpurun_out/r03_h_2m128_bb$bb.err
  echo "== 2M x 128, rounds of $bb: $(grep 'ehx build' gpurun_out/r03_h_2m128_bb$bb.err | tail -1)"; python scripts/jl.py gpurun_out/r03_h_2m128_bb$bb.jsonl | cut -c1-140
done
