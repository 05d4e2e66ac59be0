#!/bin/bash
# kernel trace of the bench workload with a given library build: per-pass scan durations
# usage: gpu_trace_lib.sh ROWS libsuffix...   (suffix "" = shipped library)
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
R=$(pwd)
ROWS=$1; shift
for sfx in "$@"; do
  L=$R/embeddinghub_amd/lib/libehx$sfx.so
  rm -rf gpurun_out/prof/t$sfx
  (cd /tmp && EHX_LIB=$L timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/prof/t$sfx -o trace -- python $R/bench.py --rows $ROWS --steps 3 --warmup 1 --no-cpu-baseline --no-f32-engine > $R/gpurun_out/prof/t$sfx.log 2>&1)
  echo "== lib '$sfx'"; python scripts/rocpd_summary.py gpurun_out/prof/t$sfx | grep -E "flat_scan|last scan" | cut -c1-150
done
