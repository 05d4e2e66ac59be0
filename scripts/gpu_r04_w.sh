#!/bin/bash
# round 4, session w (numbers of the final code): full GPU suite, smoke(), rocprofv3 passes of the headline workload
# (trace with launch sequence, FETCH / WRITE, SQ, clock), then the default bench run
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
R=$(pwd); O=$R/gpurun_out
( timeout 900 python -m pytest tests -m gpu -x -q --timeout=600 2>&1 | tail -8 ) > $O/r04_w_pytest_gpu_tail.txt; tail -3 $O/r04_w_pytest_gpu_tail.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee -a $O/r04_w_pytest_gpu_tail.txt
TAG=r04_w BENCH_ARGS="--config-legs 0" bash scripts/gpu_profile_i8.sh 2>&1 | tail -8
cp $O/prof/r04_w_*summary.txt $O/prof/r04_w_i8_traffic.json $O/ 2>/dev/null
timeout 600 python bench.py --steps 20 --warmup 5 > $O/r04_w_bench_default.json 2> $O/r04_w_bench_default_progress.txt; echo "bench rc=$?"
tail -16 $O/r04_w_bench_default_progress.txt
python - <<'PY'
import json
try:
    r = json.loads(open("gpurun_out/r04_w_bench_default.json").read().strip().splitlines()[-1])
    print("value", r["value"], "ms", r["ms_per_step"], "roof", r["roofline"]["frac"], r["roofline"]["kernel_ms"])
    for k in ("device_resident_queries", "host_pointer_one_caller"):
        print(k, r.get(k))
    print("exactness", r.get("exactness"))
    for n, leg in (r.get("configs") or {}).items():
        print(n, leg.get("value"), leg.get("ms_per_step"), leg.get("roofline", {}).get("frac"), leg.get("fallback_queries"), (leg.get("exactness") or {}).get("ids_identical_to_oracle"), (leg.get("exactness") or {}).get("oracle_rows"))
    print("single_query", json.dumps(r.get("single_query"))[:600])
    gs = r.get("graph_path_structured") or {}
    print("structured op", gs.get("operating_point"), gs.get("exact_flat_engine_same_rows_queries_per_s"), gs.get("exact_flat_engine_fallback_queries"))
    gp = r.get("graph_path") or {}
    print("graph", gp.get("operating_point"), gp.get("roofline"), gp.get("hbm_bytes_of_the_graph_space"), gp.get("rows_fp32_bytes"))
    print("set_concurrent", r.get("set_concurrent"))
    print("skipped", r.get("optional_legs_skipped"))
except Exception as e:
    print("parse failed", e)
PY
find $O/prof -name "*.db" -size +4M -delete
