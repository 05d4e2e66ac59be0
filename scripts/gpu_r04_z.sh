#!/bin/bash
# round 4, session z: graph suites after "a bulk build releases its scratch", HBM held by a 4 M x 768 graph space
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
( timeout 900 python -m pytest tests/test_graph_parity.py tests/test_fuzz_graph.py tests/test_graph_scale.py tests/test_concurrent_set.py -m gpu -x -q --timeout=800 2>&1 | tail -5 ) > $O/r04_z_pytest_tail.txt; tail -3 $O/r04_z_pytest_tail.txt
: > $O/r04_z_graph_footprint.jsonl
timeout 300 python scripts/graph_footprint.py --rows 4000000 --dims 768 2>/dev/null | tail -1 | tee -a $O/r04_z_graph_footprint.jsonl
EHX_GRAPH_TWO_COPIES=1 timeout 300 python scripts/graph_footprint.py --rows 4000000 --dims 768 2>/dev/null | tail -1 | tee -a $O/r04_z_graph_footprint.jsonl
