#!/bin/bash
# round 4, last session: short rows with the query tile resident in LDS (QRES, k_flati8.hip) — the flat parity suites on
# it (default on), then same-box A/B against EHX_I8_QRES=0
export TMPDIR=/tmp
mkdir -p gpurun_out
R=$(pwd); O=$R/gpurun_out
( timeout 600 python -m pytest tests/test_i8_filter.py tests/test_flat_parity.py tests/test_exactness.py tests/test_fuzz_parity.py -m gpu -x -q --timeout=400 2>&1 | tail -8 ) > $O/r04_qres_pytest_tail.txt; tail -3 $O/r04_qres_pytest_tail.txt
: > $O/r04_qres_ab_flat.jsonl
for shape in "--rows 6250000 --dims 128 --metric l2" "--rows 1000000 --dims 128" "--rows 2000000 --dims 192 --metric ip" "--rows 2000000 --dims 64 --metric l2"; do
  for f in 0 1 0 1; do
    EHX_I8_QRES=$f timeout 120 python scripts/ab_flat.py $shape --label "qres=$f" 2>$O/ab_err.txt | tail -1 >> $O/r04_qres_ab_flat.jsonl || tail -5 $O/ab_err.txt
  done
done
python - <<'PY'
import json
for l in open("gpurun_out/r04_qres_ab_flat.jsonl"):
    r = json.loads(l)
    print(r["label"], r["rows"], r["dims"], r["metric"], "ms", r["ms_per_step"], "kernel", r["kernel_ms"], r["engine"], "fb", r["i8_fallback"], r["filter_fallback"], r["exhaustive"], "chk", r["ids_checksum_last_batch"])
PY
