#!/bin/bash
# round 3, session d: int8 scan ablations; GPU-built vs oracle-built graphs at two round shares; the new shard tests
mkdir -p gpurun_out
export TMPDIR=/tmp
./scripts/ablate_i8.sh 10000000 "" _abl1 _abl2 _abl4 _abl5 _abl13 > gpurun_out/r03_d_ablate_i8.txt 2>&1; cat gpurun_out/r03_d_ablate_i8.txt
for div in 16 64; do
  EHX_BUILD_DIV=$div EHX_SCALE_REPORT=$PWD/gpurun_out/r03_d_scale_report.jsonl timeout 600 python -m pytest tests/test_graph_scale.py -m gpu -q -k "cos20k768 or cos200k768" > gpurun_out/r03_d_scale_div$div.log 2>&1; echo "scale div=$div rc=$?"; tail -3 gpurun_out/r03_d_scale_div$div.log
done
cat gpurun_out/r03_d_scale_report.jsonl
timeout 600 python -m pytest tests/test_shards_abi.py tests/test_graph_scale.py -m gpu -q -k "large_k or dropping or ties or larger_than_one_dispatch" > gpurun_out/r03_d_new_tests.log 2>&1; echo "new tests rc=$?"; tail -15 gpurun_out/r03_d_new_tests.log
