#!/bin/bash
# round 4, session x: the headline from host pointers with long-lived caller threads, per-call durations in the line
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
ARGS="--steps 20 --warmup 5 --check-queries 0 --no-f32-engine --graph-rows 0 --structured-rows 0 --set-concurrent 0 --single-query 0 --no-cpu-baseline --config-legs 0"
: > $O/r04_x_bench_short.jsonl
for i in 1 2; do
  timeout 300 python bench.py $ARGS 2>$O/r04_x_err.txt | tail -1 >> $O/r04_x_bench_short.jsonl || tail -5 $O/r04_x_err.txt
done
python - <<'PY'
import json
for l in open("gpurun_out/r04_x_bench_short.jsonl"):
    r = json.loads(l)
    print("host", r["ms_per_step"], "kernel", r["roofline"]["kernel_ms"], "device", r["device_resident_queries"]["ms_per_step"], r["device_resident_queries"]["kernel_ms"], "one caller", r["host_pointer_one_caller"]["ms_per_step"])
    print("  calls", r["config"]["host_calls"])
PY
