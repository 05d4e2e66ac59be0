#!/bin/bash
# round 4, session e: uniform |A| per lane group (exact integer alarm) — parity first, then clocks/cycles A/B
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
( timeout 600 python -m pytest tests/test_flat_parity.py tests/test_i8_filter.py tests/test_exactness.py tests/test_fuzz_parity.py tests/test_concurrent_set.py -m gpu -x -q --timeout=600 2>&1 | tail -8 ) > $O/r04_e_pytest_tail.txt; tail -4 $O/r04_e_pytest_tail.txt
TAG=r04_e LIBS="_r03 - _abl2" bash scripts/gpu_r04_d.sh
TAG=r04_e_1250k LIBS="_r03 -" ROWS=1250000 bash scripts/gpu_r04_d.sh
TAG=r04_e_6250k128 LIBS="_r03 -" ROWS=6250000 DIMS=128 METRIC=l2 bash scripts/gpu_r04_d.sh
