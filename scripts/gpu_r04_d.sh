#!/bin/bash
# round 4, session d: ablations WITH the shader clock of each run (kernel trace + GRBM_GUI_ACTIVE in one pass): the part
# is power-managed, so a build that skips work may run at a different clock — cycles and milliseconds are both reported.
# EHX_I8_ABL bits: 2 phase 1 only (no slow path), 4 no DMA after the prologue, 8 no fragment reads
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
R=$(pwd)
O=$R/gpurun_out
OUT=$O/${TAG:-r04_d}_i8_ablations_clk.txt
: > $OUT
for sfx in ${LIBS:-_r03 "" _abl2 _abl6 _abl10 _abl14}; do
  [ "$sfx" = "-" ] && sfx=""
  L=$R/embeddinghub_amd/lib/libehx$sfx.so
  rm -rf $O/prof/clk$sfx
  (cd /tmp && EHX_LIB=$L timeout 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d $O/prof/clk$sfx -o p -- python $R/scripts/ab_flat.py --rows ${ROWS:-10000000} --dims ${DIMS:-768} --metric ${METRIC:-cosine} --steps 6 --warmup 2 > $O/prof/clk$sfx.log 2>&1)
  python - "$O/prof/clk$sfx" "$sfx" >> $OUT <<'PY'
import sqlite3, sys, glob
db = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0]
c = sqlite3.connect(db)
rows = c.execute("""select d.start, d.end - d.start, (select sum(e.value) from rocpd_pmc_event e where e.event_id = d.event_id)
                    from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id
                    where s.kernel_name like '%flat_scan_i8_kernel<false%' or s.kernel_name like '%scan_i8_kernelILb0%'
                    order by d.start""").fetchall()
# the passes of the last batch (the longest dispatch is the main pass)
tail = rows[-4:]
out = []
for _, dur, cyc in tail:
    ghz = (cyc / 8.0) / dur if cyc and dur else 0.0   # 8 XCDs report their own active cycles
    out.append("%.3f ms @ %.2f GHz" % (dur / 1e6, ghz))
main = max(rows[-8:], key=lambda r: r[1])
print("lib '%s': last passes: %s | main pass %.3f ms = %.2f M cycles" % (sys.argv[2], "; ".join(out), main[1] / 1e6, (main[2] or 0) / 8e6))
PY
done
cat $OUT
find $O/prof -name "*.db" -size +4M -delete
