#!/bin/bash
# graph-path measurement suite (profiles/r0N_*_graph_*): bench_graph on every dataset of DESIGN.md's table, the phase
# timers (ablation build), a rocprofv3 kernel trace of one bench_graph run, an FETCH_SIZE pass of the same.
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
R=$(pwd)
J() { python scripts/jl.py $1 | cut -c1-150; }
timeout 200 python scripts/bench_graph.py --rows 1000000 --dims 128 --metric l2 --gpu-build --efs 50,200,800 > gpurun_out/graph_1m128.jsonl 2> gpurun_out/g1.err; J gpurun_out/graph_1m128.jsonl
timeout 200 python scripts/bench_graph.py --rows 300000 --dims 768 --metric cosine --gpu-build --efs 100,800 > gpurun_out/graph_300k768.jsonl 2> gpurun_out/g2.err; J gpurun_out/graph_300k768.jsonl
timeout 300 python scripts/bench_graph.py --rows 2000000 --dims 768 --metric cosine --gpu-build --efs 100,400,1600 --reps 3 > gpurun_out/graph_2m768.jsonl 2> gpurun_out/g3.err; J gpurun_out/graph_2m768.jsonl
timeout 300 python scripts/bench_graph.py --rows 1000000 --dims 768 --metric cosine --data manifold:32 --efs 10,20,40,80 --reps 5 > gpurun_out/graph_manifold32_1m768.jsonl 2> gpurun_out/g5.err; J gpurun_out/graph_manifold32_1m768.jsonl
timeout 300 python scripts/bench_graph.py --rows 1000000 --dims 128 --metric l2 --data manifold:16 --efs 10,20,40,80 --reps 5 > gpurun_out/graph_manifold16_1m128.jsonl 2> gpurun_out/g6.err; J gpurun_out/graph_manifold16_1m128.jsonl
bash scripts/gpu_graph_profile.sh
unset EHX_LIB
echo "== kernel trace"
rm -rf gpurun_out/prof/gtrace
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof/gtrace -o trace -- python $R/scripts/bench_graph.py --rows 300000 --dims 768 --metric cosine --gpu-build --efs 800 --reps 10 > $R/gpurun_out/prof/gtrace.log 2>&1); tail -1 gpurun_out/prof/gtrace.log | cut -c1-300
python scripts/rocpd_summary.py gpurun_out/prof/gtrace > gpurun_out/prof/gtrace_summary.txt 2>&1; head -12 gpurun_out/prof/gtrace_summary.txt | cut -c1-170
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/prof/gpmc_$c
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/prof/gpmc_$c -o pmc -- python $R/scripts/bench_graph.py --rows 300000 --dims 768 --metric cosine --gpu-build --efs 800 --reps 3 > $R/gpurun_out/prof/gpmc_$c.log 2>&1)
  python scripts/rocpd_summary.py gpurun_out/prof/gpmc_$c > gpurun_out/prof/gpmc_${c}_summary.txt 2>&1; grep -h "graph_search" gpurun_out/prof/gpmc_${c}_summary.txt | cut -c1-170
done
find gpurun_out/prof -name "*.db" -size +20M -delete
echo "== 6.25M x 128 (one of eight shards of configs[4])"
timeout 400 python scripts/bench_graph.py --rows 6250000 --dims 128 --metric l2 --gpu-build --efs 50,200,800 --reps 5 > gpurun_out/graph_6250k128.jsonl 2> gpurun_out/g7.err; J gpurun_out/graph_6250k128.jsonl
echo "== 10M x 768"
timeout 500 python scripts/bench_graph.py --rows 10000000 --dims 768 --metric cosine --gpu-build --efs 100,400,1600 --reps 3 > gpurun_out/graph_10m768.jsonl 2> gpurun_out/g4.err; J gpurun_out/graph_10m768.jsonl
