#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python tests/test_fuzz_parity.py 4000 11 2>&1 | tail -12 | cut -c1-600 | tee gpurun_out/r03_y_fuzz.txt
