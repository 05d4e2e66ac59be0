#!/bin/bash
# round 5, session i: graph mode's one-launch single query (parity first, then BASELINE configs[0] latency with the
# launch count switched by EHX_ONE_LAUNCH), and the cascade's first pass / growth on the bench shapes
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
timeout 900 python -m pytest tests/test_graph_parity.py tests/test_flat_parity.py -x -q -m gpu -k "one_query or single_query or strict_parity" 2>&1 | tail -5 | tee $O/r05_i_pytest_tail.txt
EHX_ONE_LAUNCH=0 timeout 300 python scripts/c1_single_query.py 2>/dev/null | tail -1 > $O/r05_i_c1_single_query_two_launches.json
timeout 300 python scripts/c1_single_query.py 2>/dev/null | tail -1 > $O/r05_i_c1_single_query_one_launch.json
python - <<'PY'
import json
for f in ("two_launches", "one_launch"):
    r = json.load(open("gpurun_out/r05_i_c1_single_query_%s.json" % f))
    print(f, [(p["path"], p.get("ef"), p["gpu_latency_us_median"], p.get("cpu_oracle_latency_us_1_thread"), p.get("queries_identical_to_oracle")) for p in r["points"]])
PY
OUT=$O/r05_i_cascade.jsonl
: > $OUT
one() {  # rows dims metric extra first growth keys
  EHX_I8_FIRST_TILES=$5 EHX_I8_GROWTH=$6 EHX_I8_FIRST_KEYS=$7 timeout 200 python scripts/ab_flat.py --rows $1 --dims $2 --metric $3 $4 --steps 60 --warmup 8 --label "first=$5 growth=$6 keys=$7" 2>/dev/null | tail -1 >> $OUT
}
for cfg in "512 4 0" "256 8 64" "256 8 0" "256 4 0" "128 8 64"; do
  one 1250000 768 cosine "" $cfg
  one 10000000 768 cosine "" $cfg
  one 6250000 128 l2 "" $cfg
done
for cfg in "512 4 0" "256 8 64"; do
  one 12500000 1536 cosine "--f16 --steps 12" $cfg
  one 200000 768 cosine "" $cfg
done
python - <<'PY'
import json
for l in open("gpurun_out/r05_i_cascade.jsonl"):
    r = json.loads(l)
    print(r["rows"], r["dims"], r["label"], "ms", r["ms_per_step"], "kernel", r["kernel_ms"], "fb", r["i8_fallback"], r["filter_fallback"], r["exhaustive"], "chk", r["ids_checksum_last_batch"])
PY
