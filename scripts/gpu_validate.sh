#!/bin/bash
# One GPU-box session: the GPU parity tests, smoke(), the default bench (one JSON line -> gpurun_out/bench_default.json).
# usage: gpurun --timeout 1300 -- bash scripts/gpu_validate.sh
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest -m gpu"; timeout 700 python -m pytest tests -m gpu -x -q --timeout=180 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest_gpu.log
echo "== smoke"; timeout 200 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/smoke.log
echo "== bench default"; timeout 500 python bench.py > gpurun_out/bench_default.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/bench_default.log > gpurun_out/bench_default.json; python - <<'PY'
import json
r = json.loads(open("gpurun_out/bench_default.json").read())
print("value", r["value"], "ms", r["ms_per_step"], "frac", r["roofline"]["frac"], "f32", r.get("f32_scan_engine", {}).get("frac"))
print("graph_path", json.dumps(r.get("graph_path"))[:900])
print("cpu", r.get("cpu_baseline", {}).get("value"))
PY
