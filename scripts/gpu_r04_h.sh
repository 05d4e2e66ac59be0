#!/bin/bash
# round 4, session h: k' = 128 floor re-check (i8 filter tests), lock-step and threshold-rank variants on the new kernel
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
R=$(pwd); O=$R/gpurun_out
( timeout 600 python -m pytest tests/test_i8_filter.py tests/test_flat_parity.py -m gpu -x -q --timeout=600 2>&1 | tail -5 ) > $O/r04_h_pytest_tail.txt; tail -3 $O/r04_h_pytest_tail.txt
: > $O/r04_h_variants.jsonl
for v in "base:" "sync:EHX_I8_SYNC=1" "safety1.5:EHX_I8_SAFETY=1.5" "safety3:EHX_I8_SAFETY=3" "growth8:EHX_I8_GROWTH=8" "growth2:EHX_I8_GROWTH=2"; do
  name=${v%%:*}; envs=${v#*:}
  for shape in "--rows 10000000 --dims 768" "--rows 1250000 --dims 768"; do
    env $envs timeout 200 python scripts/ab_flat.py $shape --label "$name" 2>$O/ab_err.txt | tail -1 >> $O/r04_h_variants.jsonl || tail -5 $O/ab_err.txt
  done
done
python - <<'PY'
import json
for l in open("gpurun_out/r04_h_variants.jsonl"):
    r = json.loads(l)
    print(r["label"], r["rows"], r["dims"], "ms", r["ms_per_step"], "kernel", r["kernel_ms"], "fb", r["i8_fallback"], r["filter_fallback"], r["exhaustive"])
PY
for v in "nosync:EHX_I8_SYNC=0" "sync:EHX_I8_SYNC=1"; do
  name=${v%%:*}; envs=${v#*:}
  rm -rf $O/prof/h_$name
  (cd /tmp && env $envs timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/prof/h_$name -o p -- python $R/scripts/ab_flat.py --rows 10000000 --dims 768 --steps 4 --warmup 2 > $O/prof/h_$name.log 2>&1)
  python scripts/rocpd_summary.py $O/prof/h_$name 2>/dev/null | grep -E "scan_i8_kernelILb0.*FETCH|scan_i8_kernelILb0ELb1EEEvNS_10ScanArgsI8E.kd +[0-9]" | cut -c1-160 | sed "s/^/$name: /"
done
find $O/prof -name "*.db" -size +4M -delete
