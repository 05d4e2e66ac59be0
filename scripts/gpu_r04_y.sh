#!/bin/bash
# round 4, session y: the default bench run on the round's final bench.py (long-lived caller threads, graph-space HBM)
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
timeout 600 python bench.py > $O/r04_y_bench_default.json 2> $O/r04_y_bench_default_progress.txt; echo "bench rc=$?"
tail -6 $O/r04_y_bench_default_progress.txt
python - <<'PY'
import json
r = json.loads(open("gpurun_out/r04_y_bench_default.json").read().strip().splitlines()[-1])
print("value", r["value"], "ms", r["ms_per_step"], "roof", r["roofline"]["frac"], r["roofline"]["kernel_ms"], "traffic", r["roofline"]["traffic"])
print("host_calls", r["config"]["host_calls"])
for k in ("device_resident_queries", "host_pointer_one_caller"):
    print(k, r.get(k))
print("exactness", r.get("exactness"))
for n, leg in (r.get("configs") or {}).items():
    print(n, leg.get("value"), leg.get("ms_per_step"), leg.get("roofline", {}).get("frac"), leg.get("fallback_queries"), (leg.get("exactness") or {}).get("ids_identical_to_oracle"), (leg.get("exactness") or {}).get("oracle_rows"))
gs = r.get("graph_path_structured") or {}
print("structured op", gs.get("operating_point"), gs.get("exact_flat_engine_same_rows_queries_per_s"))
gp = r.get("graph_path") or {}
print("graph", gp.get("operating_point"), gp.get("roofline", {}).get("frac"), gp.get("hbm_bytes_of_the_graph_space"), gp.get("rows_fp32_bytes"))
print("cpu_baseline", json.dumps(r.get("cpu_baseline"))[:300])
print("skipped", r.get("optional_legs_skipped"))
PY
