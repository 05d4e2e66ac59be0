#!/bin/bash
# round 5 (prepared at the end of round 4, not run yet): the cascade's shape on a small shard — first pass size and
# growth factor of the int8 scan at 1.25 M x 768 (the 8-GPU strong-scaling shard; round 4: 1.14 ms per batch, target 1.10)
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
: > $O/r05_b_cascade_1250k.jsonl
for ft in 512 256 1024 2048; do
  for gr in 4 2 8 16; do
    EHX_I8_FIRST_TILES=$ft EHX_I8_GROWTH=$gr timeout 120 python scripts/ab_flat.py --rows 1250000 --dims 768 --steps 40 --warmup 8 --label "first=$ft growth=$gr" 2>/dev/null | tail -1 >> $O/r05_b_cascade_1250k.jsonl
  done
done
python - <<'PY'
import json
for l in open("gpurun_out/r05_b_cascade_1250k.jsonl"):
    r = json.loads(l)
    print(r["label"], "ms", r["ms_per_step"], "kernel", r["kernel_ms"], "fb", r["i8_fallback"], r["filter_fallback"], r["exhaustive"], "chk", r["ids_checksum_last_batch"])
PY
