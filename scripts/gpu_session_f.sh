#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 800 python -m pytest tests -m gpu -q --timeout=300 -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest_gpu.log
# graph bench at 1M x 128 (C4 shard shape) and 2M x 768: visit log instead of memset
python scripts/bench_graph.py --rows 1000000 --dims 128 --metric l2 --efs 200 --gpu-build --reps 5 > gpurun_out/graph_1m128.jsonl 2>gpurun_out/graph_1m128.err; tail -2 gpurun_out/graph_1m128.jsonl | cut -c1-600
python scripts/bench_graph.py --rows 2000000 --dims 768 --metric cosine --efs 100,400 --gpu-build --reps 5 > gpurun_out/graph_2m768.jsonl 2>gpurun_out/graph_2m768.err; tail -3 gpurun_out/graph_2m768.jsonl | cut -c1-600
# concurrent set at the bench shape (1M rows)
timeout 600 python bench.py --rows 1000000 --graph-rows 0 --structured-rows 0 --no-cpu-baseline --no-f32-engine --check-queries 0 --set-concurrent 262144 > gpurun_out/bench_setconc.log 2>&1
python - <<'PY'
import json
r = json.loads(open("gpurun_out/bench_setconc.log").read().strip().splitlines()[-1])
print("1M value", r["value"], "ms", r["ms_per_step"]); print(json.dumps(r.get("set_concurrent"))[:900])
PY
