#!/bin/bash
# round 5, session d: session c found the "slow mode" = ONE stall of ~60 ms inside the ten timed batches, only with a graph
# space built beside the flat one and only on the NULL stream.  Which call stalls, and in which HIP API?
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
O=gpurun_out
R=$(pwd)
: > $O/r05_d_modes.jsonl
for i in 1 2; do
  timeout 200 python scripts/studies/structured_flat_mode.py --graph-rows 131072 --batches 30 --label "graph beside #$i" 2>$O/r05_d_err.txt >> $O/r05_d_modes.jsonl || tail -5 $O/r05_d_err.txt
done
rm -rf $O/prof/r05_d_hip
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --hip-trace --output-format csv -d $R/$O/prof/r05_d_hip -o p -- python $R/scripts/studies/structured_flat_mode.py --graph-rows 131072 --batches 30 --label "graph beside, hip trace" >> $R/$O/r05_d_modes.jsonl 2>$R/$O/prof/r05_d_hip.log)
python - <<'PY'
import json, csv, glob, os
for l in open("gpurun_out/r05_d_modes.jsonl"):
    r = json.loads(l)
    print(r["label"], "|", r["phase"], "| qps", r["qps"], "wall", r["wall_ms_per_batch"], "scan", r["scan_ms_mean"], "slowest", r["slowest_call"], r["slowest_call_ms_then_sync_ms"])
files = glob.glob("gpurun_out/prof/r05_d_hip/**/*hip_api_trace.csv", recursive=True)
print(files)
for f in files:
    rows = list(csv.DictReader(open(f)))
    print(len(rows), "hip api calls; columns", list(rows[0].keys()) if rows else None)
    def dur(r):
        return (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    t_end = max(int(r["End_Timestamp"]) for r in rows)
    tail = [r for r in rows if int(r["Start_Timestamp"]) > t_end - 2_000_000_000]   # the last 2 s: the three phases
    big = sorted(tail, key=dur, reverse=True)[:25]
    for r in big:
        print("%-40s %9.3f ms  at -%.1f ms" % (r["Function"], dur(r), (t_end - int(r["Start_Timestamp"])) / 1e6))
    # around the biggest stall: the 12 calls before and after
    if big:
        b = big[0]
        i = rows.index(b)
        print("--- around the longest call ---")
        for r in rows[max(0, i - 14): i + 6]:
            print("  %-40s %9.3f ms" % (r["Function"], dur(r)))
for f in glob.glob("gpurun_out/prof/r05_d_hip/**/*kernel_trace.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    t_end = int(rows[-1]["End_Timestamp"])
    prev = None
    print("--- kernel gaps > 2 ms in the last 2 s ---")
    for r in rows:
        if int(r["Start_Timestamp"]) > t_end - 2_000_000_000 and prev is not None:
            gap = (int(r["Start_Timestamp"]) - int(prev["End_Timestamp"])) / 1e6
            if gap > 2:
                print("  gap %.2f ms before %s (after %s)" % (gap, r["Kernel_Name"][:60], prev["Kernel_Name"][:60]))
        prev = r
PY
