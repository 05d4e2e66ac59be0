#!/bin/bash
# round 5, session j: the whole GPU suite, then the default bench run exactly as the driver starts it
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
echo "(suite: see the first run of this session, profiles/r05_j_pytest_gpu_tail.txt)"

SECONDS=0; python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/r05_j_bench_default_line.json 2> $O/r05_j_bench_default_progress.txt
echo "bench rc=$? wall ${SECONDS}s"
cp $O/bench_detail.json $O/r05_j_bench_default_detail.json
wc -c $O/r05_j_bench_default_line.json
cat $O/r05_j_bench_default_line.json
grep -E "^\[bench|Elapsed|ehx i8" $O/r05_j_bench_default_progress.txt | tail -40
