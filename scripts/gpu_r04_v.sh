#!/bin/bash
# round 4, session v: the int8 scan with the tile epilogue inside the next tile's first stage (EHX_I8_FUSED) — parity
# suites of the flat path on the fused kernel, then same-box A/B, fused (default) against EHX_I8_FUSED=0 (run time switch)
export TMPDIR=/tmp
mkdir -p gpurun_out
R=$(pwd); O=$R/gpurun_out
( timeout 900 python -m pytest tests/test_i8_filter.py tests/test_flat_parity.py tests/test_exactness.py tests/test_fuzz_parity.py -m gpu -x -q --timeout=600 2>&1 | tail -8 ) > $O/r04_v_pytest_tail.txt; tail -3 $O/r04_v_pytest_tail.txt
: > $O/r04_v_ab_flat.jsonl
for shape in "--rows 10000000 --dims 768" "--rows 1250000 --dims 768" "--rows 6250000 --dims 128 --metric l2" "--rows 1000000 --dims 128" "--rows 4000000 --dims 384"; do
  for f in 0 1 0 1; do
    EHX_I8_FUSED=$f timeout 200 python scripts/ab_flat.py $shape --label "fused=$f" 2>$O/ab_err.txt | tail -1 >> $O/r04_v_ab_flat.jsonl || tail -5 $O/ab_err.txt
  done
done
python - <<'PY'
import json
for l in open("gpurun_out/r04_v_ab_flat.jsonl"):
    r = json.loads(l)
    print(r["label"], r["rows"], r["dims"], "ms", r["ms_per_step"], "kernel", r["kernel_ms"], "fb", r["i8_fallback"], r["filter_fallback"], r["exhaustive"], "chk", r["ids_checksum_last_batch"])
PY
