#!/bin/bash
# round 5, session e: the one-off ~68 ms stall sits inside a hipLaunchKernel of a scan pass, ~10 batches after a graph
# space was built beside the flat one (session d).  Which condition moves it?
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
: > $O/r05_e_modes.jsonl
run() {  # label, env, args...
  local label=$1 envs=$2; shift 2
  env $envs timeout 200 python scripts/studies/structured_flat_mode.py --graph-rows 131072 --batches 30 --label "$label" "$@" 2>$O/r05_e_err.txt >> $O/r05_e_modes.jsonl || tail -3 $O/r05_e_err.txt
}
run "baseline" "X=1"
run "own stream first" "X=1" --own-first
run "sleep 3 s after the build" "X=1" --sleep 3
run "build keeps its scratch" "EHX_BUILD_SCRATCH_KEEP=100000000000"
run "no async scratch reclaim" "HSA_ENABLE_SCRATCH_ASYNC_RECLAIM=0"
run "graph rows 16384 only" "X=1" --graph-rows 16384
python - <<'PY'
import json
for l in open("gpurun_out/r05_e_modes.jsonl"):
    r = json.loads(l)
    print(r["label"], "|", r["phase"], "| qps", r["qps"], "wall", r["wall_ms_per_batch"], "slowest", r["slowest_call"], r["slowest_call_ms_then_sync_ms"])
PY
