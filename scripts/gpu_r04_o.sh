#!/bin/bash
# round 4, session o: full GPU suite on the pipelined host path, host callers again, short bench (headline ratio)
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
( timeout 900 python -m pytest tests -m gpu -x -q --timeout=600 2>&1 | tail -6 ) > $O/r04_o_pytest_gpu_tail.txt; tail -3 $O/r04_o_pytest_gpu_tail.txt
: > $O/r04_o_host_callers.jsonl
for rows in 10000000 1250000; do
  for pipe in 0 1; do
    EHX_HOST_PIPELINE=$pipe timeout 300 python scripts/host_callers.py --rows $rows 2>$O/hc_err.txt | tail -1 | tee -a $O/r04_o_host_callers.jsonl || tail -5 $O/hc_err.txt
  done
done
for i in 1 2; do
timeout 300 python bench.py --steps 20 --warmup 5 --graph-rows 0 --structured-rows 0 --no-cpu-baseline --single-query 0 --set-concurrent 0 --config-legs 0 --check-queries 0 2> /dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('bench $i: value', r['value'], 'ms', r['ms_per_step'], 'device', r['device_resident_queries']['ms_per_step'], 'ratio', r['device_resident_queries']['host_pointer_over_device_resident'], 'one caller', r['host_pointer_one_caller']['ms_per_step'], 'roof', r['roofline']['frac'])
"
done
