#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python bench.py --rows 2000000 --dims 1536 --rows-dtype f16 --set-concurrent 200000 --steps 3 --warmup 1 --no-f32-engine --graph-rows 0 --no-cpu-baseline > gpurun_out/bench_c5_setconc.log 2>&1; echo "rc=$?"; tail -1 gpurun_out/bench_c5_setconc.log | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_step'], json.dumps(r.get('set_concurrent')))"
timeout 300 python bench.py --rows 2000000 --set-concurrent 300000 --steps 3 --warmup 1 --no-f32-engine --graph-rows 0 --no-cpu-baseline > gpurun_out/bench_c3_setconc.log 2>&1; echo "rc=$?"; tail -1 gpurun_out/bench_c3_setconc.log | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_step'], json.dumps(r.get('set_concurrent')))"
tail -n 3 gpurun_out/bench_c5_setconc.log | cut -c1-300
