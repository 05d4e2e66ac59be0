#!/bin/bash
# round 3, session m: what a SMALL build round costs (kernel durations at P = 64 .. 512): structured and Gaussian rows
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
R=$(pwd)
for data in gauss manifold:32; do
  tag=$(echo $data | tr -d ':')
  rm -rf gpurun_out/prof/r03_m_$tag
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/prof/r03_m_$tag -o trace -- python $R/scripts/bench_graph.py --rows 120000 --dims 768 --metric cosine --gpu-build --data $data --efs 20 > $R/gpurun_out/prof/r03_m_$tag.log 2>&1)
  echo "== $data"; python scripts/rocpd_summary.py gpurun_out/prof/r03_m_$tag | grep -E "insert_|kernel  " | cut -c1-150
done
