#!/bin/bash
# round 5, session g: the ~70 ms stall inside hipLaunchKernel comes and goes between processes (sessions c-e: 5 of 11 runs
# with a graph space beside the flat one, 0 of 6 without).  ROCclr's own log (AMD_LOG_LEVEL=4, microsecond timestamps)
# around the stall: what does the launch wait for?
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
: > $O/r05_g_modes.jsonl
for i in 1 2 3 4; do
  rm -f /tmp/amdlog_$i.txt
  AMD_LOG_LEVEL=4 AMD_LOG_LEVEL_FILE=/tmp/amdlog_$i.txt timeout 300 python scripts/studies/structured_flat_mode.py --graph-rows 16384 --batches 30 --label "amd log #$i" 2>$O/r05_g_err.txt >> $O/r05_g_modes.jsonl || tail -3 $O/r05_g_err.txt
  ls -la /tmp/amdlog_$i.txt* 2>/dev/null | head -3
done
python - <<'PY'
import json, re, glob
for l in open("gpurun_out/r05_g_modes.jsonl"):
    r = json.loads(l)
    print(r["label"], "|", r["phase"], "| qps", r["qps"], "wall", r["wall_ms_per_batch"], "slowest", r["slowest_call"], r["slowest_call_ms_then_sync_ms"])
pat = re.compile(r"(\d{6,}) us")
for f in sorted(glob.glob("/tmp/amdlog_*")):
    lines = open(f, errors="replace").read().splitlines()
    ts = []
    for i, l in enumerate(lines):
        m = pat.search(l)
        if m:
            ts.append((int(m.group(1)), i))
    if not ts:
        print(f, "no timestamps; first lines:", lines[:3])
        continue
    t_end = ts[-1][0]
    gaps = []
    for (t0, i0), (t1, i1) in zip(ts, ts[1:]):
        if t1 - t0 > 20000 and t0 > t_end - 4_000_000:
            gaps.append((t1 - t0, i0, i1))
    print("==", f, len(lines), "lines; gaps > 20 ms in the last 4 s:", [(g[0] / 1000.0) for g in gaps])
    for g, i0, i1 in gaps[:3]:
        print("--- gap %.1f ms ---" % (g / 1000.0))
        for l in lines[max(0, i0 - 25): i1 + 12]:
            print("   ", l[:230])
PY
