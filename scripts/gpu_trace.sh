#!/bin/bash
# kernel trace of the bench workload: per-kernel durations (rocpd sqlite -> scripts/rocpd_summary.py)
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
R=$(pwd)
ROWS=${1:-10000000}
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof/trace -o trace -- python $R/bench.py --rows $ROWS --steps 5 --warmup 1 --no-cpu-baseline --no-f32-engine > $R/gpurun_out/prof/trace.log 2>&1); tail -1 gpurun_out/prof/trace.log
python scripts/rocpd_summary.py gpurun_out/prof/trace > gpurun_out/prof/trace_summary.txt 2>&1; cat gpurun_out/prof/trace_summary.txt | head -40
