#!/bin/bash
# round 5, session l: the lock-step's cost was an artefact — the snapshot of the siblings' progress was read through a
# volatile generic pointer, which the compiler turned into flat_load_dword + s_waitcnt vmcnt(0): the wave's whole DMA
# look-ahead drained at every look (by tile: +7.7 %, by revolution: +60 %).  With a real ds_read: time and FETCH_SIZE again.
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
O=gpurun_out
timeout 600 python -m pytest tests/test_structured_rows.py tests/test_i8_filter.py -x -q -m gpu 2>&1 | tail -3 | tee $O/r05_l_pytest_tail.txt
OUT=$O/r05_l_sync.jsonl
: > $OUT
ab() {  # label env rows dims metric extra
  env $2 timeout 200 python scripts/ab_flat.py --rows $3 --dims $4 --metric $5 $6 --steps 60 --warmup 8 --label "$1" 2>/dev/null | tail -1 >> $OUT
}
for sy in 0 2 1 3 4 rev 0 2; do
  ab "sync=$sy" EHX_I8_SYNC=$sy 10000000 768 cosine ""
done
for sy in 0 2 0 2; do
  ab "sync=$sy" EHX_I8_SYNC=$sy 1250000 768 cosine ""
done
for sy in 0 2; do
  ab "sync=$sy" EHX_I8_SYNC=$sy 12500000 1536 cosine "--f16 --steps 12"
  ab "sync=$sy" EHX_I8_SYNC=$sy 4000000 768 cosine ""
done
python - <<'PY'
import json
for l in open("gpurun_out/r05_l_sync.jsonl"):
    r = json.loads(l)
    print(r["rows"], r["dims"], r["label"], "ms", r["ms_per_step"], "kernel", r["kernel_ms"], "fb", r["i8_fallback"], r["filter_fallback"], r["exhaustive"], "chk", r["ids_checksum_last_batch"])
PY
TAG=r05_l_sync2 SYNC=2 PASSES="fetch" BENCH_ARGS="--config-legs 0" bash scripts/gpu_profile_i8.sh 2>&1 | tail -9
cp gpurun_out/prof/r05_l*summary.txt gpurun_out/prof/r05_l*_i8_traffic.json gpurun_out/ 2>/dev/null
find gpurun_out/prof -name "*.db" -size +4M -delete
