#!/bin/bash
# round 4, session l: the one-launch single-query path — parity, then latency (scripts/c1_single_query.py) before / after
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
( timeout 900 python -m pytest tests/test_flat_parity.py tests/test_abi.py tests/test_exactness.py tests/test_grpc_shim.py tests/test_cpp_dropin.py -m gpu -x -q --timeout=600 2>&1 | tail -12 ) > $O/r04_l_pytest_tail.txt; tail -6 $O/r04_l_pytest_tail.txt
EHX_ONE_LAUNCH=0 timeout 200 python scripts/c1_single_query.py > $O/r04_l_c1_single_query_three_launches.json 2>$O/c1_err.txt || tail -5 $O/c1_err.txt
timeout 200 python scripts/c1_single_query.py > $O/r04_l_c1_single_query_one_launch.json 2>$O/c1_err.txt || tail -5 $O/c1_err.txt
python - <<'PY'
import json
for f in ("three_launches", "one_launch"):
    try:
        r = json.loads(open("gpurun_out/r04_l_c1_single_query_%s.json" % f).read().strip().splitlines()[-1])
        print(f, json.dumps(r)[:1200])
    except Exception as e:
        print(f, "parse failed", e)
PY
