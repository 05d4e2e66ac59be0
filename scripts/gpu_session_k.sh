#!/bin/bash
# M = 32 GPU insertion parity + the concurrency measurements with stream-level syncs + the default bench line
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_graph_parity.py tests/test_concurrent_set.py -m gpu -q -s --timeout=300 > gpurun_out/pytest_k.log 2>&1; echo "pytest rc=$?"; grep -a "search alone\|passed\|failed\|Error" gpurun_out/pytest_k.log | tail -8
timeout 900 python bench.py > gpurun_out/bench_default.log 2>&1; echo "bench rc=$?"
tail -1 gpurun_out/bench_default.log > gpurun_out/bench_default.json
python - <<'PY'
import json
r = json.loads(open("gpurun_out/bench_default.json").read())
for k in ("value", "ms_per_step", "roofline", "exactness", "set_concurrent", "host_pointer_path", "cpu_baseline"):
    print(k, json.dumps(r.get(k))[:600])
print("graph", json.dumps(r.get("graph_leg"))[:1500])
PY
