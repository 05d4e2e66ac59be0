#!/bin/bash
# rocprofv3 passes for the bench workload: kernel trace, then PMC passes (each its own run, kernel-trace
# only).  Summaries (text) land in gpurun_out/prof/*.txt; copy the ones to keep into profiles/.
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
R=$(pwd)
ARGS="--steps 5 --warmup 1 --no-cpu-baseline --no-f32-engine $BENCH_ARGS"
echo "== bench (default flags)"; timeout 900 python bench.py > gpurun_out/bench_default.log 2>&1; tail -1 gpurun_out/bench_default.log
echo "== kernel trace"
rm -rf gpurun_out/prof/trace
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof/trace -o trace -- python $R/bench.py $ARGS > $R/gpurun_out/prof/trace.log 2>&1); tail -1 gpurun_out/prof/trace.log | cut -c1-300
python scripts/rocpd_summary.py gpurun_out/prof/trace > gpurun_out/prof/trace_summary.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  echo "== pmc $c"
  rm -rf gpurun_out/prof/pmc_$c
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/prof/pmc_$c -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-f32-engine $BENCH_ARGS > $R/gpurun_out/prof/pmc_$c.log 2>&1)
  python scripts/rocpd_summary.py gpurun_out/prof/pmc_$c > gpurun_out/prof/pmc_${c}_summary.txt 2>&1
done
echo "== pmc SQ"
rm -rf gpurun_out/prof/pmc_sq
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE -d $R/gpurun_out/prof/pmc_sq -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-f32-engine $BENCH_ARGS > $R/gpurun_out/prof/pmc_sq.log 2>&1)
python scripts/rocpd_summary.py gpurun_out/prof/pmc_sq > gpurun_out/prof/pmc_sq_summary.txt 2>&1
head -30 gpurun_out/prof/trace_summary.txt | cut -c1-170
grep -h "flat_scan" gpurun_out/prof/pmc_*_summary.txt | cut -c1-170
find gpurun_out/prof -name "*.db" -size +20M -delete; du -sh gpurun_out/prof
