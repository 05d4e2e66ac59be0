#!/bin/bash
# round 5, session k: rocprofv3 passes of the final code — headline workload (trace + FETCH / WRITE / SQ / clock), the
# same workload's FETCH_SIZE with the lock-step by tile (tolerance 2), and launch-by-launch traces of the small-shard and
# short-row shapes (fractions recomputed from a rocprof trace of those shapes, not from HIP events)
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
timeout 600 python -m pytest tests/test_i8_filter.py tests/test_fuzz_parity.py tests/test_flat_parity.py -x -q -m gpu 2>&1 | tail -3 | tee gpurun_out/r05_k_pytest_tail.txt
TAG=r05_k BENCH_ARGS="--config-legs 0" bash scripts/gpu_profile_i8.sh 2>&1 | tail -24
TAG=r05_k_sync2 SYNC=2 PASSES="fetch" BENCH_ARGS="--config-legs 0" bash scripts/gpu_profile_i8.sh 2>&1 | tail -8
TAG=r05_k_1250k PASSES="trace" BENCH_ARGS="--config-legs 0 --rows 1250000 --steps 8" bash scripts/gpu_profile_i8.sh 2>&1 | tail -3
TAG=r05_k_1m PASSES="trace" BENCH_ARGS="--config-legs 0 --rows 1000000 --steps 8" bash scripts/gpu_profile_i8.sh 2>&1 | tail -3
TAG=r05_k_6250k128 PASSES="trace" BENCH_ARGS="--config-legs 0 --rows 6250000 --dims 128 --metric-kind l2 --steps 8" bash scripts/gpu_profile_i8.sh 2>&1 | tail -3
for t in r05_k_1250k r05_k_1m r05_k_6250k128; do echo "== $t"; grep -h "scan_i8_kernel\|last scan-kernel" gpurun_out/prof/${t}_i8_trace_summary.txt | cut -c1-170; done
cp gpurun_out/prof/r05_k*summary.txt gpurun_out/prof/r05_k*_i8_traffic.json gpurun_out/ 2>/dev/null
find gpurun_out/prof -name "*.db" -size +4M -delete
