#!/bin/bash
# Profiling aid: builds ablated copies of the scan kernel (EHX_ABL bits: 1 no epilogue, 2 no DMA,
# 4 no LDS fragment reads, 8 no stage barrier) and times each on the bench workload.  Results are
# wrong by construction; only kernel_ms is meaningful.
set -e
for abl in "$@"; do
  EHX_LIB_SUFFIX=_abl$abl EHX_DEFS="-DEHX_ABL=$abl" python embeddinghub_amd/build.py --force > /dev/null 2>&1
done
