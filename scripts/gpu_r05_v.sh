#!/bin/bash
# round 5, session v: PMC passes of the short-row shape (6.25 M x 128 L2) on the final library: what the half-tile
# kernel's time is made of (SQ counters, shader clock) and its HBM traffic
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
TAG=r05_v_6250k128 PASSES="sq clk fetch" BENCH_ARGS="--config-legs 0 --rows 6250000 --dims 128 --metric-kind l2 --steps 8" bash scripts/gpu_profile_i8.sh 2>&1 | tail -4
grep -h "ScanArgsI8E" gpurun_out/prof/r05_v_6250k128_i8_pmc_sq_summary.txt gpurun_out/prof/r05_v_6250k128_i8_pmc_clk_summary.txt gpurun_out/prof/r05_v_6250k128_i8_pmc_fetch_summary.txt | grep -v "^#" | cut -c1-170
cp gpurun_out/prof/r05_v*summary.txt gpurun_out/ 2>/dev/null
find gpurun_out/prof -name "*.db" -size +4M -delete
