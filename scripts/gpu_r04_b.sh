#!/bin/bash
# round 4, session b: the 16x16x64 int8 scan — parity suite, same-box A/B against round 3's library, host-pointer overlap
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
( timeout 900 python -m pytest tests -m gpu -x -q --timeout=600 2>&1 | tail -15 ) > $O/r04_b_pytest_gpu_tail.txt; tail -6 $O/r04_b_pytest_gpu_tail.txt
: > $O/r04_b_ab_flat.jsonl
for shape in "--rows 10000000 --dims 768" "--rows 1250000 --dims 768" "--rows 6250000 --dims 128 --metric l2" "--rows 1000000 --dims 128" "--rows 4000000 --dims 384"; do
  for lib in embeddinghub_amd/lib/libehx_r03.so embeddinghub_amd/lib/libehx.so; do
    EHX_LIB=$lib timeout 200 python scripts/ab_flat.py $shape --label "$(basename $lib)" 2>$O/ab_err.txt | tail -1 >> $O/r04_b_ab_flat.jsonl || tail -5 $O/ab_err.txt
  done
done
python - <<'PY'
import json
for l in open("gpurun_out/r04_b_ab_flat.jsonl"):
    r = json.loads(l)
    print(r["label"], r["rows"], r["dims"], "ms", r["ms_per_step"], "kernel", r["kernel_ms"], r["engine"], "fb", r["i8_fallback"], r["filter_fallback"], r["exhaustive"], r["ids_checksum_last_batch"])
PY
timeout 300 python bench.py --steps 20 --warmup 5 --graph-rows 0 --structured-rows 0 --no-cpu-baseline --single-query 0 --set-concurrent 0 --config-legs 0 > $O/r04_b_bench_short.json 2> $O/r04_b_bench_short_progress.txt; echo "bench rc=$?"
python - <<'PY'
import json
try:
    r = json.loads(open("gpurun_out/r04_b_bench_short.json").read().strip().splitlines()[-1])
    print("value", r["value"], "ms", r["ms_per_step"], "roof", r["roofline"]["frac"], r["roofline"]["kernel_ms"])
    for k in ("device_resident_queries", "host_pointer_one_caller"):
        print(k, r.get(k))
    print("exactness", r.get("exactness"))
except Exception as e:
    print("parse failed", e); print(open("gpurun_out/r04_b_bench_short_progress.txt").read()[-2000:])
PY
