#!/bin/bash
# round 4, session p: short rows — how the candidate-list length k' moves the scan (EHX_I8_KPRIME), fallbacks counted
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
: > $O/r04_p_kprime_short_rows.jsonl
for shape in "--rows 6250000 --dims 128 --metric l2" "--rows 1000000 --dims 128" "--rows 4000000 --dims 384" "--rows 10000000 --dims 256"; do
  for kp in 0 64 128; do
    EHX_I8_KPRIME=$kp timeout 200 python scripts/ab_flat.py $shape --steps 40 --warmup 5 --label "kprime=$kp" 2>$O/ab_err.txt | tail -1 >> $O/r04_p_kprime_short_rows.jsonl || tail -5 $O/ab_err.txt
  done
done
python - <<'PY'
import json
for l in open("gpurun_out/r04_p_kprime_short_rows.jsonl"):
    r = json.loads(l)
    print(r["label"], r["rows"], r["dims"], "ms", r["ms_per_step"], "kernel", r["kernel_ms"], "fb", r["i8_fallback"], r["filter_fallback"], r["exhaustive"])
PY
