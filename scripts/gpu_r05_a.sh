#!/bin/bash
# round 5, first session (prepared at the end of round 4, not run yet): why is the exact engine's rate on the structured
# leg's rows bimodal (0.12-0.14 M and 1.05-1.10 M queries/s across round 4's default runs)?  The structured
# leg alone, with the flag pairs of the slow and of the fast runs, the int8 engine's adaptation trace in the progress log
# (bench.py sets EHX_I8_TRACE=1) and `exact_flat_engine_detail` in the line.
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
COMMON="--check-queries 0 --no-f32-engine --graph-rows 0 --set-concurrent 0 --single-query 0 --no-cpu-baseline --config-legs 0 --cpu-hnsw-seconds 0"
: > $O/r05_a_structured.jsonl
for flags in "--steps 20 --warmup 5" "--steps 10 --warmup 2" "--steps 20 --warmup 5" "--steps 10 --warmup 2"; do
  timeout 300 python bench.py $flags $COMMON 2> $O/r05_a_progress_last.txt | tail -1 >> $O/r05_a_lines.jsonl
  python -c "import json;print(json.dumps(json.load(open('gpurun_out/bench_detail.json'))))" >> $O/r05_a_structured.jsonl
  grep -h "ehx i8\|structured" $O/r05_a_progress_last.txt | tail -12
done
python - <<'PY'
import json
for l in open("gpurun_out/r05_a_structured.jsonl"):
    r = json.loads(l)
    gs = r.get("graph_path_structured") or {}
    print(r["steps"], r["warmup"], gs.get("exact_flat_engine_same_rows_queries_per_s"), gs.get("exact_flat_engine_detail"))
PY
