#!/bin/bash
# round 4, session g: full GPU suite on the 16x16x64 kernel, rocprofv3 passes of the headline workload (trace, FETCH /
# WRITE for profiles/traffic.json, SQ, clock), then the default bench run
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
( timeout 900 python -m pytest tests -m gpu -x -q --timeout=600 2>&1 | tail -8 ) > $O/r04_g_pytest_gpu_tail.txt; tail -3 $O/r04_g_pytest_gpu_tail.txt
TAG=r04_g BENCH_ARGS="--config-legs 0" bash scripts/gpu_profile_i8.sh 2>&1 | tail -45
cp gpurun_out/prof/r04_g_*summary.txt gpurun_out/prof/r04_g_i8_traffic.json $O/ 2>/dev/null
timeout 500 python bench.py > $O/r04_g_bench_default.json 2> $O/r04_g_bench_default_progress.txt; echo "bench rc=$?"
tail -12 $O/r04_g_bench_default_progress.txt
python - <<'PY'
import json
try:
    r = json.loads(open("gpurun_out/r04_g_bench_default.json").read().strip().splitlines()[-1])
    print("value", r["value"], "ms", r["ms_per_step"], "roof", r["roofline"]["frac"], r["roofline"]["kernel_ms"])
    for k in ("device_resident_queries", "host_pointer_one_caller"):
        print(k, r.get(k))
    print("exactness", r.get("exactness"))
    for n, leg in (r.get("configs") or {}).items():
        print(n, leg.get("value"), leg.get("ms_per_step"), leg.get("roofline", {}).get("frac"), leg.get("fallback_queries"), (leg.get("exactness") or {}).get("ids_identical_to_oracle"), (leg.get("exactness") or {}).get("oracle_rows"))
    print("skipped", r.get("optional_legs_skipped"))
except Exception as e:
    print("parse failed", e)
PY
