#!/bin/bash
# round 4, session m (final numbers): full GPU suite, rocprofv3 passes of the headline workload on the final kernel
# (trace with launch sequence, FETCH / WRITE, SQ, clock), launch-by-launch trace of a 1.25 M x 768 batch, same-box A/B
# against round 3's library, then the default bench run and smoke()
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
R=$(pwd); O=$R/gpurun_out
( timeout 900 python -m pytest tests -m gpu -x -q --timeout=600 2>&1 | tail -8 ) > $O/r04_m_pytest_gpu_tail.txt; tail -3 $O/r04_m_pytest_gpu_tail.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee -a $O/r04_m_pytest_gpu_tail.txt
TAG=r04_m BENCH_ARGS="--config-legs 0" bash scripts/gpu_profile_i8.sh 2>&1 | tail -8
cp $O/prof/r04_m_*summary.txt $O/prof/r04_m_i8_traffic.json $O/ 2>/dev/null
rm -rf $O/prof/m_1250k
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $O/prof/m_1250k -o p -- python $R/scripts/ab_flat.py --rows 1250000 --dims 768 --steps 6 --warmup 2 > $O/prof/m_1250k.log 2>&1)
ROCPD_SEQ=16 python scripts/rocpd_summary.py $O/prof/m_1250k > $O/r04_m_trace_1250k_rows_summary.txt 2>&1
: > $O/r04_m_ab_flat.jsonl
for shape in "--rows 10000000 --dims 768" "--rows 1250000 --dims 768" "--rows 1000000 --dims 768" "--rows 6250000 --dims 128 --metric l2" "--rows 1000000 --dims 128" "--rows 4000000 --dims 384" "--rows 12500000 --dims 1536 --f16 --steps 6 --warmup 3"; do
  for lib in embeddinghub_amd/lib/libehx_r03.so embeddinghub_amd/lib/libehx.so; do
    EHX_LIB=$lib timeout 200 python scripts/ab_flat.py $shape --label "$(basename $lib)" 2>$O/ab_err.txt | tail -1 >> $O/r04_m_ab_flat.jsonl || tail -5 $O/ab_err.txt
  done
done
python - <<'PY'
import json
for l in open("gpurun_out/r04_m_ab_flat.jsonl"):
    r = json.loads(l)
    print(r["label"], r["rows"], r["dims"], "ms", r["ms_per_step"], "kernel", r["kernel_ms"], "fb", r["i8_fallback"], r["filter_fallback"], r["exhaustive"])
PY
timeout 500 python bench.py --steps 20 --warmup 5 > $O/r04_m_bench_default.json 2> $O/r04_m_bench_default_progress.txt; echo "bench rc=$?"
tail -14 $O/r04_m_bench_default_progress.txt
python - <<'PY'
import json
try:
    r = json.loads(open("gpurun_out/r04_m_bench_default.json").read().strip().splitlines()[-1])
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
    print("graph", gp.get("operating_point"), gp.get("roofline"))
    print("set_concurrent", r.get("set_concurrent"))
    print("skipped", r.get("optional_legs_skipped"))
except Exception as e:
    print("parse failed", e)
PY
find $O/prof -name "*.db" -size +4M -delete
