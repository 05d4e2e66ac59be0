#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_fuzz_parity.py tests/test_fuzz_graph.py tests/test_flat_parity.py -x -q -m gpu 2>&1 | tail -3
timeout 200 python bench.py --rows 1000000 --steps 40 --warmup 5 --no-cpu-baseline --no-f32-engine --graph-rows 0 --structured-rows 0 --set-concurrent 0 --check-queries 64 2>/dev/null | tail -1 > gpurun_out/r03_p_1m.json
python - <<'P'
import json
r = json.load(open("gpurun_out/r03_p_1m.json"))
print("1M:", r["ms_per_step"], r["value"], r["exactness"]["ids_identical_to_oracle"], json.dumps(r["single_query"])[:600])
P
