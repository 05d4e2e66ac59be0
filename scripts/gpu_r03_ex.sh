#!/bin/bash
# the RCCL exchange step on a 1-GPU box: the one-rank GPU test, then the 1.25 M-row shard (8-GPU strong-scaling share of
# the headline workload) without and with the exchange step of the multi-rank path (world of one rank)
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_sharded.py tests/test_shards_abi.py -x -q -m gpu 2>&1 | tail -6
F="--rows 1250000 --steps 40 --warmup 5 --no-cpu-baseline --no-f32-engine --graph-rows 0 --structured-rows 0 --set-concurrent 0 --single-query 0 --check-queries 64"
for tag in plain exchange plain2 exchange2; do
  X=""; case $tag in exchange*) X="--exchange";; esac
  timeout 300 python bench.py $F $X 2>gpurun_out/ex_$tag.err | tail -1 > gpurun_out/r03_ex_$tag.json
  python - $tag <<'P'
import json, sys
r = json.load(open("gpurun_out/r03_ex_%s.json" % sys.argv[1]))
print(sys.argv[1], r["ms_per_step"], r["value"], r["config"]["parallelism"][:60], r.get("exactness", {}).get("ids_identical_to_oracle"))
P
done
