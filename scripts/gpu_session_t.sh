#!/bin/bash
# cascade growth factor at shard sizes (strong scaling over 8 GPUs leaves 1.25 M rows per GPU)
mkdir -p gpurun_out
export TMPDIR=/tmp
S="--graph-rows 0 --structured-rows 0 --no-cpu-baseline --no-f32-engine --check-queries 0"
for rows in 1250000 2500000 5000000; do
for g in 4 8 16 1000; do
  name=r${rows}g$g
  EHX_I8_GROWTH=$g timeout 300 python bench.py $S --rows $rows > gpurun_out/bench_$name.log 2>&1
  tail -1 gpurun_out/bench_$name.log > gpurun_out/bench_$name.json
  python - <<PY
import json
try:
    r = json.loads(open("gpurun_out/bench_$name.json").read())
    print("$name", "value", r["value"], "ms", r["ms_per_step"], "kernel_ms", r["roofline"]["kernel_ms"], "i8fb", r.get("i8_fallback_queries"))
except Exception as e:
    print("$name parse failed", e); print(open("gpurun_out/bench_$name.log").read()[-800:])
PY
done
done
