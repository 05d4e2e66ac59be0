#!/bin/bash
# round 5, session m: lock-step by tile with the progress store issued BEFORE the tile's epilogue (session l: +5.3 % with
# the store after it, whatever the tolerance — a fixed cost: the out-of-order store makes the next counted wait stricter)
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
O=gpurun_out
OUT=$O/r05_m_sync.jsonl
: > $OUT
ab() {  # label env rows dims metric extra
  env $2 timeout 200 python scripts/ab_flat.py --rows $3 --dims $4 --metric $5 $6 --steps 60 --warmup 8 --label "$1" 2>/dev/null | tail -1 >> $OUT
}
for sy in 0 2 3 0 2; do
  ab "sync=$sy" EHX_I8_SYNC=$sy 10000000 768 cosine ""
done
for sy in 0 2 0 2; do
  ab "sync=$sy" EHX_I8_SYNC=$sy 1250000 768 cosine ""
done
for sy in 0 2; do
  ab "sync=$sy" EHX_I8_SYNC=$sy 4000000 768 cosine ""
  ab "sync=$sy" EHX_I8_SYNC=$sy 12500000 1536 cosine "--f16 --steps 12"
done
python - <<'PY'
import json
for l in open("gpurun_out/r05_m_sync.jsonl"):
    r = json.loads(l)
    print(r["rows"], r["dims"], r["label"], "ms", r["ms_per_step"], "kernel", r["kernel_ms"], "fb", r["i8_fallback"], r["filter_fallback"], r["exhaustive"], "chk", r["ids_checksum_last_batch"])
PY
