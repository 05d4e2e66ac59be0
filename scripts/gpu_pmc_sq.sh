#!/bin/bash
# SQ stall-category counters of the scan kernels on the bench workload (own run: --pmc with kernel-trace only)
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
R=$(pwd)
ROWS=${1:-10000000}
rm -rf gpurun_out/prof/pmc_sq
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS -d $R/gpurun_out/prof/pmc_sq -o sq -- python $R/bench.py --rows $ROWS --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof/pmc_sq.log 2>&1)
tail -1 gpurun_out/prof/pmc_sq.log | cut -c1-200
python scripts/rocpd_summary.py gpurun_out/prof/pmc_sq | grep -E "flat_scan|counter" | cut -c1-160
rm -rf gpurun_out/prof/pmc_sq2
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_LDS_IDX_ACTIVE -d $R/gpurun_out/prof/pmc_sq2 -o sq -- python $R/bench.py --rows $ROWS --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof/pmc_sq2.log 2>&1)
tail -1 gpurun_out/prof/pmc_sq2.log | cut -c1-200
python scripts/rocpd_summary.py gpurun_out/prof/pmc_sq2 | grep -E "flat_scan|counter" | cut -c1-160
