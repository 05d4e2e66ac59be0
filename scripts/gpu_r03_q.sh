#!/bin/bash
# same-box A/B of the query preparation: libehx_base.so (three launches before the first scan) vs libehx.so (one)
export TMPDIR=/tmp
mkdir -p gpurun_out
R=$(pwd)
timeout 300 python -m pytest tests/test_fuzz_parity.py tests/test_flat_parity.py tests/test_i8_filter.py -x -q -m gpu 2>&1 | tail -3
F="--rows 1000000 --steps 100 --warmup 10 --no-cpu-baseline --no-f32-engine --graph-rows 0 --structured-rows 0 --set-concurrent 0 --single-query 0 --check-queries 64"
for tag in old new old2 new2; do
  L=$R/embeddinghub_amd/lib/libehx.so; case $tag in old*) L=$R/embeddinghub_amd/lib/libehx_base.so;; esac
  EHX_LIB=$L timeout 200 python bench.py $F 2>/dev/null | tail -1 > gpurun_out/r03_q_$tag.json
  python - $tag <<'P'
import json, sys
r = json.load(open("gpurun_out/r03_q_%s.json" % sys.argv[1]))
print(json.dumps({"run": sys.argv[1], "rows": 1000000, "ms_per_step": r["ms_per_step"], "value": r["value"], "kernel_ms": r["roofline"]["kernel_ms"], "ids_identical_to_oracle": r["exactness"]["ids_identical_to_oracle"]}))
P
done | tee gpurun_out/r03_q_prep_ab.jsonl
