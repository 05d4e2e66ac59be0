#!/bin/bash
# times ablated builds of the fp16 filter scan on the bench workload (results are wrong by construction;
# only kernel_ms is meaningful).  usage: ablate16.sh ROWS abl...
mkdir -p gpurun_out
ROWS=$1; shift
for abl in "$@"; do
  L=embeddinghub_amd/lib/libehx_abl$abl.so
  [ "$abl" = 0 ] && L=embeddinghub_amd/lib/libehx.so
  echo -n "abl=$abl  "
  EHX_LIB=$PWD/$L timeout 300 python bench.py --rows $ROWS --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('ms_per_step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'])"
done
