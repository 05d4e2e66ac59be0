#!/bin/bash
# round 4, session q: single-copy graph storage — every graph test, the i8 tests with k' by row length, graph bench A/B
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
( timeout 1200 python -m pytest tests/test_graph_parity.py tests/test_fuzz_graph.py tests/test_graph_scale.py tests/test_i8_filter.py tests/test_shards_abi.py tests/test_go_conformance.py tests/test_grpc_shim.py -m gpu -x -q --timeout=800 2>&1 | tail -12 ) > $O/r04_q_pytest_tail.txt; tail -8 $O/r04_q_pytest_tail.txt
for two in 1 0; do
  EHX_GRAPH_TWO_COPIES=$two timeout 400 python scripts/bench_graph.py --rows 2000000 --dims 768 --metric cosine --gpu-build --build-batch 4096 --efs 100,400 --batches 1024 --reps 5 2>/dev/null | tee $O/r04_q_graph_2m768_twocopies$two.jsonl | python -c "
import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: continue
    print('two_copies=$two', r['workload'][:90], 'kernel_ms', r['kernel_ms'], 'recall', r['recall_at_k'], 'n_dist', r['n_dist_per_query'], 'frac', r['roofline']['frac'])
"
  EHX_GRAPH_TWO_COPIES=$two timeout 300 python scripts/bench_graph.py --rows 2000000 --dims 128 --metric l2 --gpu-build --build-batch 4096 --efs 200 --batches 1024 --reps 5 2>/dev/null | tee $O/r04_q_graph_2m128_twocopies$two.jsonl | python -c "
import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: continue
    print('two_copies=$two', r['workload'][:90], 'kernel_ms', r['kernel_ms'], 'recall', r['recall_at_k'], 'n_dist', r['n_dist_per_query'], 'frac', r['roofline']['frac'])
"
done
