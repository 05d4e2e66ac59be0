#!/bin/bash
# kernel trace of a short headline bench run: per-kernel durations + the launch sequence of the last batches
# usage: gpurun -- bash scripts/gpu_trace.sh [NAME] [bench args...]   (env for the run: TRACE_ENV="A=1 B=2")
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
R=$(pwd)
NAME=${1:-trace}; shift
ARGS="--steps 3 --warmup 1 --check-queries 0 --no-f32-engine --graph-rows 0 --structured-rows 0 --no-cpu-baseline $@"
rm -rf gpurun_out/prof/$NAME
(cd /tmp && env $TRACE_ENV timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof/$NAME -o trace -- python $R/bench.py $ARGS > $R/gpurun_out/prof/$NAME.log 2>&1)
tail -1 gpurun_out/prof/$NAME.log | cut -c1-400
ROCPD_SEQ=${SEQ:-40} python scripts/rocpd_summary.py gpurun_out/prof/$NAME > gpurun_out/prof/${NAME}_summary.txt 2>&1
cut -c1-170 gpurun_out/prof/${NAME}_summary.txt | head -80
find gpurun_out/prof -name "*.db" -size +20M -delete
