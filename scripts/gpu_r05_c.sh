#!/bin/bash
# round 5, session c: the slow mode of the exact engine on structured rows is OUTSIDE the scan phase (session a).  The flat
# space alone, several processes (the mode was constant per process), wall time beside the engine's event times; then one
# process under rocprofv3 --kernel-trace for the launch-by-launch picture.
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
O=gpurun_out
R=$(pwd)
: > $O/r05_c_modes.jsonl
for i in 1 2 3; do
  timeout 200 python scripts/studies/structured_flat_mode.py --label "flat only #$i" 2>$O/r05_c_err.txt >> $O/r05_c_modes.jsonl || tail -5 $O/r05_c_err.txt
done
timeout 200 python scripts/studies/structured_flat_mode.py --graph-rows 131072 --label "graph space beside" 2>>$O/r05_c_err.txt >> $O/r05_c_modes.jsonl
timeout 200 python scripts/studies/structured_flat_mode.py --gauss --label "gaussian control" 2>>$O/r05_c_err.txt >> $O/r05_c_modes.jsonl
for i in 1 2; do
  rm -rf $O/prof/r05_c_trace$i
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof/r05_c_trace$i -o p -- python $R/scripts/studies/structured_flat_mode.py --label "under rocprofv3 #$i" >> $R/$O/r05_c_modes.jsonl 2>$R/$O/prof/r05_c_trace$i.log)
  ROCPD_SEQ=60 python scripts/rocpd_summary.py $O/prof/r05_c_trace$i > $O/prof/r05_c_trace${i}_summary.txt 2>&1
done
cat $O/r05_c_modes.jsonl | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r['label'], '|', r['phase'], '| qps', r['qps'], 'wall', r['wall_ms_per_batch'], 'scan', r['scan_ms_mean'], 'total', r['last_total_ms'], 'fb', r['fallbacks'], r['call_ms_then_sync_ms'][:3])
"
tail -70 $O/prof/r05_c_trace1_summary.txt
