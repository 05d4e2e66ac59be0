#!/bin/bash
# round 4, session i: integer alarm levels / shared K — i8 tests, then same-box A/B against round 3 and a launch-by-launch
# trace of one 1.25 M x 768 batch
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
R=$(pwd); O=$R/gpurun_out
( timeout 600 python -m pytest tests/test_i8_filter.py tests/test_flat_parity.py tests/test_exactness.py -m gpu -x -q --timeout=600 2>&1 | tail -5 ) > $O/r04_i_pytest_tail.txt; tail -3 $O/r04_i_pytest_tail.txt
: > $O/r04_i_ab_flat.jsonl
for shape in "--rows 10000000 --dims 768" "--rows 1250000 --dims 768" "--rows 6250000 --dims 128 --metric l2" "--rows 1000000 --dims 128" "--rows 12500000 --dims 1536 --f16 --steps 6 --warmup 3"; do
  for lib in embeddinghub_amd/lib/libehx_r03.so embeddinghub_amd/lib/libehx.so; do
    EHX_LIB=$lib timeout 200 python scripts/ab_flat.py $shape --label "$(basename $lib)" 2>$O/ab_err.txt | tail -1 >> $O/r04_i_ab_flat.jsonl || tail -5 $O/ab_err.txt
  done
done
python - <<'PY'
import json
for l in open("gpurun_out/r04_i_ab_flat.jsonl"):
    r = json.loads(l)
    print(r["label"], r["rows"], r["dims"], "ms", r["ms_per_step"], "kernel", r["kernel_ms"], "fb", r["i8_fallback"], r["filter_fallback"], r["exhaustive"], r["ids_checksum_last_batch"])
PY
rm -rf $O/prof/i_1250k
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $O/prof/i_1250k -o p -- python $R/scripts/ab_flat.py --rows 1250000 --dims 768 --steps 6 --warmup 2 > $O/prof/i_1250k.log 2>&1)
ROCPD_SEQ=16 python scripts/rocpd_summary.py $O/prof/i_1250k > $O/r04_i_trace_1250k_rows_summary.txt 2>&1; tail -18 $O/r04_i_trace_1250k_rows_summary.txt | cut -c1-120
find $O/prof -name "*.db" -size +4M -delete
