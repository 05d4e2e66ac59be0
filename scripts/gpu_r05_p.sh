#!/bin/bash
# round 5, session p: the default bench run once more with the quota protections of bench.py in place (session n's run
# was frozen by the CPU quota inside the structured leg's ten timed batches again), then graph mode on the configs[3]
# shard on round 5's code with a rocprofv3 trace (VERDICT r04 missing #6)
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
O=gpurun_out; R=$(pwd)
SECONDS=0; python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/r05_p_bench_default_line.json 2> $O/r05_p_bench_default_progress.txt
echo "bench rc=$? wall ${SECONDS}s"
cp $O/bench_detail.json $O/r05_p_bench_default_detail.json
wc -c $O/r05_p_bench_default_line.json
cat $O/r05_p_bench_default_line.json
grep -E "^\[bench|ehx i8" $O/r05_p_bench_default_progress.txt | tail -8
rm -rf $O/prof/r05_p_graph
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof/r05_p_graph -o p -- python $R/scripts/bench_graph.py --rows 6250000 --dims 128 --metric l2 --gpu-build --build-batch 4096 --efs 50,200,800 --batches 1024,2048 --reps 5 > $R/$O/r05_p_graph_6250k128.jsonl 2> $R/$O/prof/r05_p_graph.log)
python scripts/rocpd_summary.py $O/prof/r05_p_graph > $O/r05_p_graph_6250k128_trace_summary.txt 2>&1
python scripts/jl.py $O/r05_p_graph_6250k128.jsonl | cut -c1-230
grep -h "graph_search" $O/r05_p_graph_6250k128_trace_summary.txt | cut -c1-150
find $O/prof -name "*.db" -size +4M -delete
