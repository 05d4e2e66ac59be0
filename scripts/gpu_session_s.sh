#!/bin/bash
# re-rank: rows loaded by the wave together (staged through LDS) vs every lane walking its own row; parity tests first
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_i8_filter.py tests/test_exactness.py tests/test_flat_parity.py tests/test_shards_abi.py -m gpu -q --timeout=300 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu.log
S="--graph-rows 0 --structured-rows 0 --no-cpu-baseline --no-f32-engine --check-queries 0"
for v in "st1:EHX_RERANK_STAGED=1:" "st0:EHX_RERANK_STAGED=0:" "st1b:EHX_RERANK_STAGED=1:" "m1st1:EHX_RERANK_STAGED=1:--rows 1000000" "m1st0:EHX_RERANK_STAGED=0:--rows 1000000"; do
  name=${v%%:*}; rest=${v#*:}; envs=${rest%%:*}; args=${rest#*:}
  env $envs timeout 300 python bench.py $S $args > gpurun_out/bench_$name.log 2>&1
  tail -1 gpurun_out/bench_$name.log > gpurun_out/bench_$name.json
  python - <<PY
import json
try:
    r = json.loads(open("gpurun_out/bench_$name.json").read())
    print("$name", "value", r["value"], "ms", r["ms_per_step"], "kernel_ms", r["roofline"]["kernel_ms"], "i8fb", r.get("i8_fallback_queries"), "host", r.get("host_pointer_path", {}).get("value"))
except Exception as e:
    print("$name parse failed", e); print(open("gpurun_out/bench_$name.log").read()[-1500:])
PY
done
