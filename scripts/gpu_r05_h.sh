#!/bin/bash
# round 5, session h: the ~65 ms stall (sessions c-g) hits whatever HIP call is in flight 5-25 ms into the first search
# phase (a launch in three runs, the device synchronisation in one).  ROCclr's log around it; and does it need the
# pageable query uploads that precede the phase?
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
: > $O/r05_h_modes.jsonl
run() {  # tag, args...
  local tag=$1; shift
  rm -f /tmp/amdlog_$tag*
  AMD_LOG_LEVEL=4 AMD_LOG_LEVEL_FILE=/tmp/amdlog_$tag timeout 300 python scripts/studies/structured_flat_mode.py --graph-rows 16384 --batches 30 --label "$tag" "$@" 2>$O/r05_h_err.txt >> $O/r05_h_modes.jsonl || tail -3 $O/r05_h_err.txt
}
run baseline
run devq --device-queries
run sleep1 --sleep 1
python - <<'PY'
import json, re, glob
for l in open("gpurun_out/r05_h_modes.jsonl"):
    r = json.loads(l)
    print(r["label"], "|", r["phase"], "| qps", r["qps"], "slowest", r["slowest_call"], r["slowest_call_ms_then_sync_ms"])
pat = re.compile(r"(\d{6,}) us")
for f in sorted(glob.glob("/tmp/amdlog_*")):
    lines = open(f, errors="replace").read().splitlines()
    ts = []
    for i, l in enumerate(lines):
        m = pat.search(l)
        if m:
            ts.append((int(m.group(1)), i))
    gaps = [(t1 - t0, i0, i1) for (t0, i0), (t1, i1) in zip(ts, ts[1:]) if 30000 < t1 - t0 < 74000]
    # the stall: a gap whose surroundings mention the int8 scan (not the host-side query generation)
    print("==", f, len(lines), "lines; gaps of 30-74 ms:", [round(g[0] / 1000.0, 1) for g in gaps])
    shown = 0
    for g, i0, i1 in gaps:
        ctx = "\n".join(lines[max(0, i0 - 60): i1 + 5])
        if "flat_scan_i8" in ctx or "select256" in ctx or "rerank256" in ctx:
            print("--- gap %.1f ms (search phase) ---" % (g / 1000.0))
            for l in lines[max(0, i0 - 14): i1 + 30]:
                print("   ", l[:200])
            shown += 1
            if shown >= 2:
                break
PY
