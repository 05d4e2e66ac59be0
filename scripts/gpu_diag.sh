#!/bin/bash
# where does the default bench spend its time?  legs one at a time, each under its own timeout, progress notes on stderr
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== structured leg only"
timeout 100 python bench.py --steps 2 --warmup 1 --graph-rows 0 --no-cpu-baseline --check-queries 0 --no-f32-engine --rows 1000000 > gpurun_out/diag_s.json 2> gpurun_out/diag_s.err; echo "rc=$?"; grep "bench " gpurun_out/diag_s.err | tail -8
echo "== graph leg only (2 M rows)"
timeout 100 python bench.py --steps 2 --warmup 1 --graph-rows 2000000 --structured-rows 0 --no-cpu-baseline --check-queries 0 --no-f32-engine --rows 2000000 > gpurun_out/diag_g.json 2> gpurun_out/diag_g.err; echo "rc=$?"; grep "bench " gpurun_out/diag_g.err | tail -8
