#!/bin/bash
# round 4, session j: helper wave of the graph search — parity suites, then same-box A/B (EHX_GRAPH_HELPER=0/1) on a
# GPU-built 2 M x 768 cosine index and a 1 M x 384 index, batch 1024 and 2048
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
( timeout 900 python -m pytest tests/test_graph_parity.py tests/test_fuzz_graph.py tests/test_graph_scale.py -m gpu -x -q --timeout=600 2>&1 | tail -6 ) > $O/r04_j_pytest_graph_tail.txt; tail -4 $O/r04_j_pytest_graph_tail.txt
for h in 0 1; do
  EHX_GRAPH_HELPER=$h timeout 400 python scripts/bench_graph.py --rows 2000000 --dims 768 --metric cosine --gpu-build --build-batch 4096 --efs 100,400 --batches 1024,2048 --reps 5 2>/dev/null | tee $O/r04_j_graph_2m768_helper$h.jsonl | python -c "
import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: continue
    print('helper $h', {k: r.get(k) for k in ('rows','dims','batch','ef','qps','kernel_ms','recall','n_dist_per_query','achieved_GBps','frac_of_hbm_peak') if k in r})
"
done
for h in 0 1; do
  EHX_GRAPH_HELPER=$h timeout 300 python scripts/bench_graph.py --rows 1000000 --dims 384 --metric cosine --gpu-build --build-batch 4096 --efs 200 --batches 1024 --reps 5 2>/dev/null | tee $O/r04_j_graph_1m384_helper$h.jsonl | python -c "
import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: continue
    print('helper $h', {k: r.get(k) for k in ('rows','dims','batch','ef','qps','kernel_ms','recall','n_dist_per_query','achieved_GBps','frac_of_hbm_peak') if k in r})
"
done
