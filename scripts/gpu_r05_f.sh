#!/bin/bash
# round 5, session f: the cascade's shape on a small shard (1 M x 768; pass 1 of 512 tiles runs at half the steady-state
# rate: its loose threshold lets ~256 keys per query through the epilogue's slow path) — first pass size, growth, and the
# number of keys the first pass aims for (EHX_I8_FIRST_KEYS, new)
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
OUT=$O/r05_f_cascade.jsonl
: > $OUT
# the stall of sessions c-h = the cgroup's CPU quota (numpy's BLAS threads burn it): the counters, and the cure
cat /sys/fs/cgroup/cpu.max 2>/dev/null; nproc
for bt in "" 8; do
  SFM_BLAS_THREADS=$bt timeout 200 python scripts/studies/structured_flat_mode.py --graph-rows 16384 --batches 30 --label "blas threads '$bt'" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r['label'], '|', r['phase'], '| qps', r['qps'], 'slowest', r['slowest_call_ms_then_sync_ms'], 'throttled (times, ms)', r['cgroup_throttled_times_ms'])
" | tee -a $O/r05_f_throttle.txt
done
# parity of the round's kernel changes first (half-tile workgroups for d <= 128, lock-step by tile): a fast wrong kernel is not done
timeout 900 python -m pytest tests/test_i8_filter.py tests/test_structured_rows.py tests/test_fuzz_parity.py -x -q -m gpu 2>&1 | tail -6 | tee $O/r05_f_pytest_tail.txt
one() {  # rows dims metric first growth keys
  EHX_I8_FIRST_TILES=$4 EHX_I8_GROWTH=$5 EHX_I8_FIRST_KEYS=$6 timeout 120 python scripts/ab_flat.py --rows $1 --dims $2 --metric $3 --steps 60 --warmup 8 --label "first=$4 growth=$5 keys=$6" 2>/dev/null | tail -1 >> $OUT
}
for cfg in "512 4 0" "512 4 128" "512 4 64" "256 4 96" "1024 4 96" "512 8 96" "256 8 64" "512 4 0"; do
  one 1000000 768 cosine $cfg
done
for cfg in "512 4 0" "512 4 64"; do
  one 1250000 768 cosine $cfg
  one 10000000 768 cosine $cfg
done
# half-tile workgroups (two per CU) against the full-tile QRES kernel, short rows
ab() {  # label env rows dims metric
  env $2 timeout 120 python scripts/ab_flat.py --rows $3 --dims $4 --metric $5 --steps 60 --warmup 8 --label "$1" 2>/dev/null | tail -1 >> $OUT
}
for rep in 1 2; do
  ab "half=0" EHX_I8_HALF=0 6250000 128 l2
  ab "half=1" EHX_I8_HALF=1 6250000 128 l2
  ab "half=0" EHX_I8_HALF=0 1000000 128 cosine
  ab "half=1" EHX_I8_HALF=1 1000000 128 cosine
done
# lock-step by tile at the headline shape: time (traffic: session h's PMC passes)
for sy in 0 1 2 3 0; do
  ab "sync=$sy" EHX_I8_SYNC=$sy 10000000 768 cosine
done
python - <<'PY'
import json
for l in open("gpurun_out/r05_f_cascade.jsonl"):
    r = json.loads(l)
    print(r["rows"], r["dims"], r["label"], "ms", r["ms_per_step"], "kernel", r["kernel_ms"], "fb", r["i8_fallback"], r["filter_fallback"], r["exhaustive"], "chk", r["ids_checksum_last_batch"])
PY
