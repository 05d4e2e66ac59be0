#!/bin/bash
# round 5, session t: the final library (one host batch enqueued as one block) — the whole GPU suite, smoke(), the default bench run as the driver starts it
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 | tee $O/r05_t_pytest_gpu_tail.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a $O/r05_t_pytest_gpu_tail.txt
SECONDS=0; python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/r05_t_bench_default_line.json 2> $O/r05_t_bench_default_progress.txt
echo "bench rc=$? wall ${SECONDS}s"
cp $O/bench_detail.json $O/r05_t_bench_default_detail.json
wc -c $O/r05_t_bench_default_line.json
cat $O/r05_t_bench_default_line.json
grep -E "^\[bench|ehx i8" $O/r05_t_bench_default_progress.txt | tail -14
