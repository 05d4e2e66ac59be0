#!/bin/bash
# Where the int8 scan's time goes: the shipped kernel next to ablation builds (k_flati8.hip, EHX_I8_ABL bits:
# 1 no epilogue, 2 phase 1 only, 4 no DMA after the prologue, 8 no fragment reads).  Build them first:
#   for v in 1 2 4 5 13; do EHX_LIB_SUFFIX=_abl$v EHX_DEFS="-DEHX_I8_ABL=$v -DEHX_ABL=1" python -m embeddinghub_amd.build; done
# usage: scripts/ablate_i8.sh ROWS [suffix ...]    ("" = the shipped library)
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
R=$(pwd)
ROWS=$1; shift
for sfx in "$@"; do
  L=$R/embeddinghub_amd/lib/libehx$sfx.so
  rm -rf gpurun_out/prof/abl$sfx
  (cd /tmp && EHX_LIB=$L timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/prof/abl$sfx -o trace -- python $R/bench.py --rows $ROWS --steps 4 --warmup 2 --no-cpu-baseline --no-f32-engine --graph-rows 0 --structured-rows 0 --set-concurrent 0 --single-query 0 --check-queries 0 > $R/gpurun_out/prof/abl$sfx.log 2>&1)
  echo "== lib '$sfx'"; python scripts/rocpd_summary.py gpurun_out/prof/abl$sfx 2>/dev/null | grep -E "flat_scan_i8|last scan" | cut -c1-170
done
