#!/bin/bash
# same-box A/B of the query preparation (canonical norm staged through LDS vs one lane reading global memory)
export TMPDIR=/tmp
mkdir -p gpurun_out
R=$(pwd)
F="--rows 1000000 --steps 100 --warmup 10 --no-cpu-baseline --no-f32-engine --graph-rows 0 --structured-rows 0 --set-concurrent 0 --single-query 0 --check-queries 0"
for tag in old new old2 new2; do
  L=$R/embeddinghub_amd/lib/libehx.so; case $tag in old*) L=$R/embeddinghub_amd/lib/libehx_oldprep.so;; esac
  EHX_LIB=$L timeout 200 python bench.py $F 2>/dev/null | tail -1 > gpurun_out/r03_q_$tag.json
  python - $tag <<'P'
import json, sys
r = json.load(open("gpurun_out/r03_q_%s.json" % sys.argv[1]))
print(json.dumps({"run": sys.argv[1], "rows": 1000000, "ms_per_step": r["ms_per_step"], "value": r["value"], "kernel_ms": r["roofline"]["kernel_ms"]}))
P
done | tee gpurun_out/r03_q_prep_ab.jsonl
