#!/bin/bash
# One GPU-box session: full GPU tests, then the headline bench on the int8 engine with and without lock-step,
# and on the fp16 filter for reference (short legs: no graph / cpu legs).
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q --timeout=300 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu.log
S="--graph-rows 0 --structured-rows 0 --no-cpu-baseline"
for v in ${AB_VARIANTS:-"i8:EHX_I8_SYNC=0:" "i8sync:EHX_I8_SYNC=1:" "i8k128:EHX_I8_KPRIME=128:" "i8g8:EHX_I8_GROWTH=8:"}; do
  name=${v%%:*}; rest=${v#*:}; envs=${rest%%:*}; args=${rest#*:}
  env $envs timeout 300 python bench.py $S $args > gpurun_out/bench_$name.log 2>&1; echo "$name rc=$?"
  tail -1 gpurun_out/bench_$name.log > gpurun_out/bench_$name.json
  python - <<PY
import json
try:
    r = json.loads(open("gpurun_out/bench_$name.json").read())
    print("$name", "value", r["value"], "ms", r["ms_per_step"], "kernel_ms", r["roofline"]["kernel_ms"], "frac", r["roofline"]["frac"], "dtype", r["dtype"], "i8fb", r.get("i8_fallback_queries"), "fallback", r["filter_fallback_queries"], "exact", json.dumps(r["exactness"])[:300])
except Exception as e:
    print("$name parse failed", e); print(open("gpurun_out/bench_$name.log").read()[-1500:])
PY
done
