#!/bin/bash
# round 4, session n: two scratch sets — concurrency / parity tests, then host callers vs device-resident, pipelined or not
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
( timeout 900 python -m pytest tests/test_flat_parity.py tests/test_i8_filter.py tests/test_concurrent_set.py tests/test_exactness.py tests/test_sharded.py tests/test_grpc_shim.py tests/test_abi.py -m gpu -x -q --timeout=600 2>&1 | tail -6 ) > $O/r04_n_pytest_tail.txt; tail -4 $O/r04_n_pytest_tail.txt
: > $O/r04_n_host_callers.jsonl
for rows in 10000000 1250000; do
  for pipe in 0 1; do
    EHX_HOST_PIPELINE=$pipe timeout 300 python scripts/host_callers.py --rows $rows 2>$O/hc_err.txt | tail -1 | tee -a $O/r04_n_host_callers.jsonl || tail -5 $O/hc_err.txt
  done
done
