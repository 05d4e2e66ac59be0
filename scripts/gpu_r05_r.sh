#!/bin/bash
# round 5, session r: two half-tile workgroups per CU run in phase (launched together, identical work): does a start skew
# of the second-resident wave of every SIMD make one's epilogue overlap the other's matrix stages?
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
OUT=$O/r05_r_skew.jsonl
: > $OUT
ab() {  # label env rows dims metric
  env $2 timeout 120 python scripts/ab_flat.py --rows $3 --dims $4 --metric $5 --steps 60 --warmup 8 --label "$1" 2>/dev/null | tail -1 >> $OUT
}
for sk in 0 16 32 48 64 96 0 48; do
  ab "skew=$sk" EHX_I8_SKEW=$sk 6250000 128 l2
done
for sk in 0 32 64; do
  ab "skew=$sk" EHX_I8_SKEW=$sk 1000000 128 cosine
done
python - <<'PY'
import json
for l in open("gpurun_out/r05_r_skew.jsonl"):
    r = json.loads(l)
    print(r["rows"], r["dims"], r["label"], "ms", r["ms_per_step"], "kernel", r["kernel_ms"], "fb", r["i8_fallback"], r["filter_fallback"], r["exhaustive"], "chk", r["ids_checksum_last_batch"])
PY
