#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
for v in 1 0 1 0; do
  EHX_GRAPH_VISLOG=$v python scripts/bench_graph.py --rows 1000000 --dims 128 --metric l2 --efs 200 --gpu-build --reps 5 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('vislog=$v 1Mx128 ef200 kernel_ms', r['kernel_ms'], 'qps_host', r['qps_host_pointers'], 'frac', r['roofline'].get('frac'))"
done
for v in 1 0; do
  EHX_GRAPH_VISLOG=$v python scripts/bench_graph.py --rows 2000000 --dims 768 --metric cosine --efs 100,400 --gpu-build --reps 5 2>/dev/null | tail -2 | python -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l); print('vislog=$v 2Mx768', r['workload'][-8:], 'kernel_ms', r['kernel_ms'], 'qps_host', r['qps_host_pointers'], 'frac', r['roofline'].get('frac'))"
done
bash scripts/gpu_profile_i8.sh
