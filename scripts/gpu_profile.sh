#!/bin/bash
# rocprofv3 passes for the bench workload: kernel trace + stats, then PMC passes (own runs).
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$(pwd)
echo "== bench (default flags)"; timeout 600 python bench.py > gpurun_out/bench_default.log 2>&1; tail -1 gpurun_out/bench_default.log
echo "== kernel trace"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof/trace -o trace -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof/trace.log 2>&1); tail -2 gpurun_out/prof/trace.log
echo "== pmc FETCH_SIZE"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/prof/pmc_fetch -o fetch -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof/pmc_fetch.log 2>&1); tail -1 gpurun_out/prof/pmc_fetch.log
echo "== pmc WRITE_SIZE"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/prof/pmc_write -o write -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof/pmc_write.log 2>&1); tail -1 gpurun_out/prof/pmc_write.log
echo "== pmc SQ"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d $R/gpurun_out/prof/pmc_sq -o sq -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof/pmc_sq.log 2>&1); tail -1 gpurun_out/prof/pmc_sq.log
find gpurun_out/prof -type f | head -50; du -sh gpurun_out/prof
