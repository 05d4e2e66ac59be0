#!/bin/bash
# round 5, session q: the default bench run on the final bench.py (untimed batches instead of sleeps before a leg's timed run)
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
SECONDS=0; python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/r05_q_bench_default_line.json 2> $O/r05_q_bench_default_progress.txt
echo "bench rc=$? wall ${SECONDS}s"
cp $O/bench_detail.json $O/r05_q_bench_default_detail.json
wc -c $O/r05_q_bench_default_line.json
cat $O/r05_q_bench_default_line.json
grep -E "^\[bench|ehx i8" $O/r05_q_bench_default_progress.txt | tail -6
