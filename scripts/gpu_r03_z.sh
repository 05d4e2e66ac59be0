#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 700 python tests/test_fuzz_graph.py 2500 6 2>&1 | tail -15 | cut -c1-900 | tee gpurun_out/r03_z_fuzz_graph.txt
