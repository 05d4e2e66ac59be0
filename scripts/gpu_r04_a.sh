#!/bin/bash
# round 4, session a: GPU suite on the new code (ordered tiles, host slots, sub-batch kernels), the MFMA-shape ubench,
# same-box A/B of the flat scan against round 3's library, then the default bench run (new headline + config legs).
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
( timeout 900 python -m pytest tests -m gpu -x -q --timeout=600 2>&1 | tail -15 ) > $O/r04_a_pytest_gpu_tail.txt; cat $O/r04_a_pytest_gpu_tail.txt | tail -4
( timeout 120 scripts/ubench/i8_mfma_shapes.bin 2>&1 ) > $O/r04_a_i8_mfma_shapes_ubench.txt; cat $O/r04_a_i8_mfma_shapes_ubench.txt
: > $O/r04_a_ab_flat.jsonl
for shape in "--rows 10000000 --dims 768" "--rows 1250000 --dims 768" "--rows 6250000 --dims 128 --metric l2" "--rows 1000000 --dims 128"; do
  for lib in embeddinghub_amd/lib/libehx_r03.so embeddinghub_amd/lib/libehx.so; do
    EHX_LIB=$lib timeout 200 python scripts/ab_flat.py $shape --label "$(basename $lib)" 2>$O/ab_err.txt | tail -1 >> $O/r04_a_ab_flat.jsonl || tail -5 $O/ab_err.txt
  done
done
cat $O/r04_a_ab_flat.jsonl | cut -c1-400
timeout 500 python bench.py > $O/r04_a_bench_default.json 2> $O/r04_a_bench_default_progress.txt; echo "bench rc=$?"
tail -25 $O/r04_a_bench_default_progress.txt
python - <<'PY'
import json
try:
    r = json.loads(open("gpurun_out/r04_a_bench_default.json").read().strip().splitlines()[-1])
    print("value", r["value"], "ms", r["ms_per_step"], "roof", r["roofline"]["frac"], r["roofline"]["kernel_ms"])
    for k in ("device_resident_queries", "host_pointer_one_caller"):
        print(k, r.get(k))
    print("exactness", r.get("exactness"))
    for n, leg in (r.get("configs") or {}).items():
        print(n, json.dumps(leg)[:700])
    gs = r.get("graph_path_structured") or {}
    print("structured", json.dumps(gs.get("recall_vs_ef"))[:1500], gs.get("exact_flat_engine_same_rows_queries_per_s"))
    print("skipped", r.get("optional_legs_skipped"))
except Exception as e:
    print("parse failed", e)
PY
