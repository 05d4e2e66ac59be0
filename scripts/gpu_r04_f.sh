#!/bin/bash
# round 4, session f: lean hit path + k' by row count — parity, then clocks/cycles A/B at three shapes, then end-to-end A/B
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
( timeout 900 python -m pytest tests -m gpu -x -q --timeout=600 2>&1 | tail -8 ) > $O/r04_f_pytest_tail.txt; tail -4 $O/r04_f_pytest_tail.txt
TAG=r04_f LIBS="_r03 - _abl2" bash scripts/gpu_r04_d.sh
TAG=r04_f_1250k LIBS="_r03 -" ROWS=1250000 bash scripts/gpu_r04_d.sh
TAG=r04_f_6250k128 LIBS="_r03 -" ROWS=6250000 DIMS=128 METRIC=l2 bash scripts/gpu_r04_d.sh
: > $O/r04_f_ab_flat.jsonl
for shape in "--rows 10000000 --dims 768" "--rows 1250000 --dims 768" "--rows 1000000 --dims 768" "--rows 6250000 --dims 128 --metric l2" "--rows 1000000 --dims 128" "--rows 4000000 --dims 384" "--rows 12500000 --dims 1536 --f16 --steps 6 --warmup 3"; do
  for lib in embeddinghub_amd/lib/libehx_r03.so embeddinghub_amd/lib/libehx.so; do
    EHX_LIB=$lib timeout 200 python scripts/ab_flat.py $shape --label "$(basename $lib)" 2>$O/ab_err.txt | tail -1 >> $O/r04_f_ab_flat.jsonl || tail -5 $O/ab_err.txt
  done
done
python - <<'PY'
import json
for l in open("gpurun_out/r04_f_ab_flat.jsonl"):
    r = json.loads(l)
    print(r["label"], r["rows"], r["dims"], "ms", r["ms_per_step"], "kernel", r["kernel_ms"], r["engine"], "fb", r["i8_fallback"], r["filter_fallback"], r["exhaustive"], r["ids_checksum_last_batch"])
PY
