#!/bin/bash
# round 4, session k: HBM traffic of graph_search_kernel on the BASELINE configs[2] shape (10 M x 768 cosine, GPU-built,
# ef 400 and 1600, batch 1024): FETCH_SIZE pass at 10 M rows, WRITE_SIZE pass at 2 M rows (writes are the visited
# bitmaps and logs: small); plus the new structured scale test
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
R=$(pwd); O=$R/gpurun_out
( timeout 900 python -m pytest tests/test_graph_scale.py -m gpu -x -q --timeout=800 2>&1 | tail -6 ) > $O/r04_k_pytest_graph_scale_tail.txt; tail -4 $O/r04_k_pytest_graph_scale_tail.txt
rm -rf $O/prof/k_fetch $O/prof/k_write
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/prof/k_fetch -o p -- python $R/scripts/bench_graph.py --rows 10000000 --dims 768 --metric cosine --gpu-build --build-batch 4096 --efs 400,1600 --batches 1024 --reps 3 > $O/r04_k_graph_10m768_pmc_fetch.jsonl 2> $O/prof/k_fetch.log)
python scripts/rocpd_summary.py $O/prof/k_fetch > $O/r04_k_graph_pmc_fetch_summary.txt 2>&1
grep -E "graph_search" $O/r04_k_graph_pmc_fetch_summary.txt | cut -c1-170
cat $O/r04_k_graph_10m768_pmc_fetch.jsonl | cut -c1-600
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/prof/k_write -o p -- python $R/scripts/bench_graph.py --rows 2000000 --dims 768 --metric cosine --gpu-build --build-batch 4096 --efs 400,1600 --batches 1024 --reps 3 > $O/r04_k_graph_2m768_pmc_write.jsonl 2> $O/prof/k_write.log)
python scripts/rocpd_summary.py $O/prof/k_write > $O/r04_k_graph_pmc_write_summary.txt 2>&1
grep -E "graph_search" $O/r04_k_graph_pmc_write_summary.txt | cut -c1-170
find $O/prof -name "*.db" -size +4M -delete
