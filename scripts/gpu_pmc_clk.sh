#!/bin/bash
# effective shader clock of the scan kernels: GRBM_GUI_ACTIVE (cycles) / dispatch duration
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
R=$(pwd)
rm -rf gpurun_out/prof/pmc_clk
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d $R/gpurun_out/prof/pmc_clk -o pmc -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline ${BENCH_ARGS:---no-f32-engine} > $R/gpurun_out/prof/pmc_clk.log 2>&1)
python - <<'PY'
import sqlite3, glob
db = sorted(glob.glob("gpurun_out/prof/pmc_clk/**/*.db", recursive=True))[0]
c = sqlite3.connect(db)
rows = c.execute("""select s.kernel_name, d.end - d.start, e.value from rocpd_pmc_event e
  join rocpd_info_pmc p on e.pmc_id = p.id join rocpd_kernel_dispatch d on d.event_id = e.event_id
  join rocpd_info_kernel_symbol s on d.kernel_id = s.id where p.name = 'GRBM_GUI_ACTIVE' and s.kernel_name like '%flat_scan%'
  order by d.start""").fetchall()
for name, ns, cyc in rows[-8:]:
    print("%-40s %9.3f ms  %12.0f cycles  -> %.3f GHz" % (name.split("(")[0][-40:], ns / 1e6, cyc, cyc / ns))
PY
