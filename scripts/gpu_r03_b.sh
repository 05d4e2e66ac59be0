#!/bin/bash
# round 3, session b: the device-paced graph build — parity suites, build rates with the trace, kernel trace of a build
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
R=$(pwd)
timeout 900 python -m pytest tests/test_graph_parity.py tests/test_go_conformance.py tests/test_grpc_shim.py tests/test_shards_abi.py tests/test_concurrent_set.py -m gpu -x -q > gpurun_out/r03_b_tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r03_b_tests.log
EHX_BUILD_TRACE=1 timeout 300 python scripts/bench_graph.py --rows 2000000 --dims 768 --metric cosine --gpu-build --efs 100,400 > gpurun_out/r03_b_graph_2m768.jsonl 2> gpurun_out/r03_b_graph_2m768.err; echo "2m768 rc=$?"; grep "ehx build" gpurun_out/r03_b_graph_2m768.err | tail -3; python scripts/jl.py gpurun_out/r03_b_graph_2m768.jsonl | cut -c1-200
EHX_BUILD_TRACE=1 timeout 300 python scripts/bench_graph.py --rows 2000000 --dims 128 --metric l2 --gpu-build --efs 50,200 > gpurun_out/r03_b_graph_2m128.jsonl 2> gpurun_out/r03_b_graph_2m128.err; echo "2m128 rc=$?"; grep "ehx build" gpurun_out/r03_b_graph_2m128.err | tail -2; python scripts/jl.py gpurun_out/r03_b_graph_2m128.jsonl | cut -c1-200
rm -rf gpurun_out/prof/r03_b_build
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof/r03_b_build -o trace -- python $R/scripts/bench_graph.py --rows 500000 --dims 768 --metric cosine --gpu-build --efs 100 > $R/gpurun_out/prof/r03_b_build.log 2>&1)
python scripts/rocpd_summary.py gpurun_out/prof/r03_b_build 2>/dev/null | head -20 | cut -c1-160
find gpurun_out/prof/r03_b_build -name "*stats*" | head; for f in $(find gpurun_out/prof/r03_b_build -name "*kernel_stats*.csv" | head -1); do head -12 $f | cut -c1-200; done
