#!/bin/bash
# round 4, session t: single-copy storage as the default for every metric — cosine A/B on one box in alternating order,
# a 128-dim pair (the two-rows-per-group path), then the graph suites under the default
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
run() {  # label, env, args
  env $2 timeout 400 python scripts/bench_graph.py $3 --gpu-build --build-batch 4096 --batches 1024 --reps 5 2>/dev/null | tee -a $O/r04_t_graph_copies.jsonl | python -c "
import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: continue
    print('$1', r['workload'][:24], r['workload'].split('insertion')[-1][:12], 'ef', r['workload'].split('ef=')[-1], 'kernel_ms', r['kernel_ms'], 'n_dist', r['n_dist_per_query'], 'frac', r['roofline']['frac'])
"
}
: > $O/r04_t_graph_copies.jsonl
run "two copies" "EHX_GRAPH_TWO_COPIES=1" "--rows 2000000 --dims 768 --metric cosine --efs 100,400"
run "one copy  " "EHX_GRAPH_TWO_COPIES=0" "--rows 2000000 --dims 768 --metric cosine --efs 100,400"
run "two copies" "EHX_GRAPH_TWO_COPIES=1" "--rows 2000000 --dims 768 --metric cosine --efs 100,400"
run "one copy  " "EHX_GRAPH_TWO_COPIES=0" "--rows 2000000 --dims 768 --metric cosine --efs 100,400"
run "two copies" "EHX_GRAPH_TWO_COPIES=1" "--rows 2000000 --dims 128 --metric cosine --efs 100,400"
run "one copy  " "EHX_GRAPH_TWO_COPIES=0" "--rows 2000000 --dims 128 --metric cosine --efs 100,400"
( timeout 1200 python -m pytest tests/test_graph_parity.py tests/test_fuzz_graph.py tests/test_graph_scale.py -m gpu -x -q --timeout=800 2>&1 | tail -6 ) > $O/r04_t_pytest_tail.txt; tail -3 $O/r04_t_pytest_tail.txt
