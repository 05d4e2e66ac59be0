#!/bin/bash
# round 5, session u: launch-by-launch rocprofv3 trace of the configs[3] shard (6.25 M x 128 L2) on the final library
# (half-tile workgroups with the start skew), for the fraction recomputed from a trace
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
TAG=r05_u_6250k128 PASSES="trace" BENCH_ARGS="--config-legs 0 --rows 6250000 --dims 128 --metric-kind l2 --steps 8" bash scripts/gpu_profile_i8.sh 2>&1 | tail -3
grep -h "ScanArgsI8E" gpurun_out/prof/r05_u_6250k128_i8_trace_summary.txt | cut -c1-170
cp gpurun_out/prof/r05_u*summary.txt gpurun_out/ 2>/dev/null
find gpurun_out/prof -name "*.db" -size +4M -delete
