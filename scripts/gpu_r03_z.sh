#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 700 python tests/test_fuzz_parity.py 1500 21 2>&1 | tail -15 | cut -c1-900 | tee gpurun_out/r03_z_fuzz_graph.txt
