#!/bin/bash
# round 4, session c: where the 16x16x64 int8 scan's time goes — ablation builds (EHX_I8_ABL bits: 2 phase 1 only,
# 4 no DMA after the prologue, 8 no fragment reads, 16 no stage barrier) next to the shipped kernel and round 3's,
# kernel trace of each on 10 M x 768 (ab_flat.py: device-resident batches, no checks — ablated results are wrong by
# construction); then SQ and clock counters of the shipped kernel.
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
R=$(pwd)
O=$R/gpurun_out
: > $O/r04_c_i8_ablations.txt
for sfx in _r03 "" _abl2 _abl4 _abl8 _abl12 _abl14 _abl30; do
  L=$R/embeddinghub_amd/lib/libehx$sfx.so
  rm -rf $O/prof/abl$sfx
  (cd /tmp && EHX_LIB=$L timeout 200 rocprofv3 --kernel-trace -d $O/prof/abl$sfx -o trace -- python $R/scripts/ab_flat.py --rows ${ROWS:-10000000} --dims ${DIMS:-768} --steps 6 --warmup 2 > $O/prof/abl$sfx.log 2>&1)
  echo "== lib '$sfx'  $(tail -1 $O/prof/abl$sfx.log | cut -c1-300)" >> $O/r04_c_i8_ablations.txt
  python scripts/rocpd_summary.py $O/prof/abl$sfx 2>/dev/null | grep -E "flat_scan_i8|last scan" | cut -c1-170 >> $O/r04_c_i8_ablations.txt
done
cat $O/r04_c_i8_ablations.txt
run_pmc() {  # name, counters...
  local name=$1; shift
  rm -rf $O/prof/$name
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $O/prof/$name -o p -- python $R/scripts/ab_flat.py --rows 10000000 --dims 768 --steps 4 --warmup 2 > $O/prof/$name.log 2>&1)
  python scripts/rocpd_summary.py $O/prof/$name > $O/${name}_summary.txt 2>&1
  grep -E "scan_i8_kernelILb0" $O/${name}_summary.txt | cut -c1-150
}
run_pmc r04_c_i8_pmc_sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE
run_pmc r04_c_i8_pmc_sq2 SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_LDS
run_pmc r04_c_i8_pmc_clk GRBM_GUI_ACTIVE
find $O/prof -name "*.db" -size +4M -delete; du -sh $O/prof
