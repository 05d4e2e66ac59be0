#!/bin/bash
# rocprofv3 passes for the headline workload on the int8 engine: kernel trace (+ launch sequence), then PMC passes
# (each its own run, kernel-trace only): FETCH_SIZE / WRITE_SIZE without and with the lock-step of sibling workgroups.
# Summaries land in gpurun_out/prof/$TAG_*.txt (TAG defaults to r03); copy the ones to keep into profiles/.
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
R=$(pwd)
ARGS="--steps 4 --warmup 1 --check-queries 0 --no-f32-engine --graph-rows 0 --structured-rows 0 --set-concurrent 0 --single-query 0 --no-cpu-baseline --reference-benchmark 0 --structured-big-rows 0 $BENCH_ARGS"
run() {  # name, env, rocprof args...
  local name=$1 envs=$2; shift 2
  rm -rf gpurun_out/prof/$name
  (cd /tmp && env $envs timeout 600 rocprofv3 --kernel-trace "$@" -d $R/gpurun_out/prof/$name -o p -- python $R/bench.py $ARGS > $R/gpurun_out/prof/$name.log 2>&1)
  ROCPD_SEQ=${SEQ:-0} python scripts/rocpd_summary.py gpurun_out/prof/$name > gpurun_out/prof/${name}_summary.txt 2>&1
}
PASSES=${PASSES:-"trace fetch write sq clk"}   # (round 5: PASSES="trace" = the launch-by-launch trace of another shape only)
SYNC=${SYNC:-0}                                # EHX_I8_SYNC of every pass (lock-step variants: SYNC=2 PASSES="fetch")
case " $PASSES " in *" trace "*) SEQ=22 run ${TAG:-r03}_i8_trace "EHX_I8_SYNC=$SYNC" --stats;; esac
case " $PASSES " in *" fetch "*) run ${TAG:-r03}_i8_pmc_fetch "EHX_I8_SYNC=$SYNC" --pmc FETCH_SIZE;; esac
case " $PASSES " in *" write "*) run ${TAG:-r03}_i8_pmc_write "EHX_I8_SYNC=$SYNC" --pmc WRITE_SIZE;; esac
case " $PASSES " in *" sq "*) run ${TAG:-r03}_i8_pmc_sq "EHX_I8_SYNC=$SYNC" --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE;; esac
case " $PASSES " in *" sq2 "*)   # what the parked / stalled cycles are made of: instruction mix and the LDS / VMEM issue buckets (names checked against the box's list)
  WANT="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_I8"
  rocprofv3 -L > gpurun_out/prof/counters_list.txt 2>&1
  HAVE=""; n=0
  for c in $WANT; do if grep -qw "$c" gpurun_out/prof/counters_list.txt && [ $n -lt 8 ]; then HAVE="$HAVE $c"; n=$((n+1)); fi; done
  echo "sq2 counters:$HAVE"
  run ${TAG:-r03}_i8_pmc_sq2 "EHX_I8_SYNC=$SYNC" --pmc $HAVE;; esac
case " $PASSES " in *" clk "*) run ${TAG:-r03}_i8_pmc_clk "EHX_I8_SYNC=$SYNC" --pmc GRBM_GUI_ACTIVE;; esac
TAG=${TAG:-r03} python - <<'PY'
import json, re
def counter_sum(path, kernel_sub, counter):
    tot, calls = 0.0, {}
    for line in open(path):
        f = line.split()
        if len(f) >= 5 and counter in f and kernel_sub in f[0]:
            tot += float(f[-1])
    return tot
def calls(path, kernel_sub):
    for line in open(path):
        f = line.split()
        if len(f) > 3 and kernel_sub in f[0] and f[1].isdigit():
            return int(f[1])
    return 0
out = {}
import os
TAG = os.environ.get("TAG", "r03")
if not os.path.exists("gpurun_out/prof/%s_i8_pmc_fetch_summary.txt" % TAG):
    raise SystemExit(0)
for tag, fpath, wpath in (("nosync", "gpurun_out/prof/%s_i8_pmc_fetch_summary.txt" % TAG, "gpurun_out/prof/%s_i8_pmc_write_summary.txt" % TAG),):
    batches = calls(fpath, "nelILb1E")   # the sample (DUMP) pass: once per batch; names are cut to their last 46 characters
    fetch = counter_sum(fpath, "ScanArgsI8E", "FETCH_SIZE")
    have_w = wpath and os.path.exists(wpath)
    write = counter_sum(wpath, "ScanArgsI8E", "WRITE_SIZE") if have_w else 0.0
    wb = calls(wpath, "nelILb1E") if have_w else 1
    per_batch = (2 * fetch / max(batches, 1) + write / max(wb, 1)) * 1024
    out[tag] = {"batches": batches, "fetch_KB_sum": fetch, "write_KB_sum": write, "bytes_per_batch": per_batch,
                "x_algorithmic_7.68e9": per_batch / (1e7 * 768 + 1024 * 768 * 4 + 1024 * 10 * 12)}
print(json.dumps(out, indent=1))
open("gpurun_out/prof/%s_i8_traffic.json" % TAG, "w").write(json.dumps(out, indent=1))
PY
head -28 gpurun_out/prof/${TAG:-r03}_i8_trace_summary.txt 2>/dev/null | cut -c1-150
grep -h "ScanArgsI8E" gpurun_out/prof/${TAG:-r03}_i8_pmc_sq_summary.txt gpurun_out/prof/${TAG:-r03}_i8_pmc_clk_summary.txt 2>/dev/null | cut -c1-140
find gpurun_out/prof -name "*.db" -size +8M -delete; du -sh gpurun_out/prof
