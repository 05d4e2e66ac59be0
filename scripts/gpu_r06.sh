#!/bin/bash
# Round 6 GPU sessions, ONE parametrised runner (no more one-file-per-session scripts):
#     gpurun --timeout N -- 'bash scripts/gpu_r06.sh <session>'
# Every session writes under gpurun_out/ (merged back by gpurun); what is to be judged is copied into profiles/r06_<session>_*.
export TMPDIR=/tmp
S=$1
O=gpurun_out
mkdir -p $O
graph_bench() {  # tag rows dims metric efs widths [extra args]
  local tag=$1 rows=$2 dims=$3 metric=$4 efs=$5 widths=$6; shift 6
  timeout 1500 python scripts/bench_graph.py --gpu-build --rows $rows --dims $dims --metric $metric --efs $efs --widths $widths "$@" \
    > $O/r06_${S}_graph_${tag}.jsonl 2> $O/r06_${S}_graph_${tag}.err
  python - <<PY
import json
for l in open("$O/r06_${S}_graph_${tag}.jsonl"):
    r = json.loads(l)
    print("%-8s ef=%-5s w=%d B=%s kern %.4f ms frac %.3f recall %.4f rows/q %.0f exp/step %.2f pf %.2f" % (
        "$tag", r["workload"].split("ef=")[1], r["search_width"], r["workload"].split("batch=")[1].split()[0], r["kernel_ms"],
        r["roofline"]["frac"], r["recall_at_k"], r["n_dist_per_query"], r["expansions_per_step"], r["prefetch_hit_rate"]))
PY
}
case "$S" in
a)  # first light of the wide walk: model parity, contract, strict parity untouched, then the two VERDICT shapes
  timeout 1200 python -m pytest tests/test_graph_wide.py -x -q -k "models_walk or contract" 2>&1 | tail -15
  timeout 900 python -m pytest tests/test_graph_parity.py tests/test_fuzz_graph.py -x -q 2>&1 | tail -5
  graph_bench 6250k128 6250000 128 l2 50,200,800 1,2,4
  graph_bench 2m768 2000000 768 cosine 100,400 1,2,4
  ;;
b)  # the recall gate at scale
  EHX_SCALE_REPORT=$O/r06_b_wide_gate.jsonl timeout 2400 python -m pytest tests/test_graph_wide.py -x -q -k "gate" -s 2>&1 | tail -15
  ;;
c)  # where a step's time goes: phase timers (-DEHX_GRAPH_PROFILE build) on the two shapes
  export EHX_LIB=$PWD/embeddinghub_amd/lib/libehx_gprof.so
  graph_bench 6250k128_gprof 6250000 128 l2 200 1,2,4
  graph_bench 2m768_gprof 2000000 768 cosine 400 1,2,4
  python - <<PY
import json, glob
for f in sorted(glob.glob("$O/r06_c_graph_*_gprof.jsonl")):
    for l in open(f):
        r = json.loads(l)
        print(f.split("graph_")[1][:12], "w=%d" % r["search_width"], "per hop" if r["phase_us_per_hop"] else "per step", r["phase_us_per_hop"] or r["phase_us_per_step"])
PY
  ;;
d)  # the wide walk with 64 rows per distance pass on short rows, entering keys compacted before the ranking, width <= ef / 8
  timeout 1200 python -m pytest tests/test_graph_wide.py -x -q -k "models_walk or contract" 2>&1 | tail -5
  graph_bench 6250k128 6250000 128 l2 50,200,800 1,2,4
  EHX_LIB=$PWD/embeddinghub_amd/lib/libehx_gprof.so graph_bench 6250k128_gprof 6250000 128 l2 200 2,4
  python - <<PY
import json
for l in open("$O/r06_d_graph_6250k128_gprof.jsonl"):
    r = json.loads(l)
    print("w=%d" % r["search_width"], "per step", r["phase_us_per_step"])
PY
  EHX_SCALE_REPORT=$O/r06_d_wide_gate.jsonl timeout 2400 python -m pytest tests/test_graph_wide.py -x -q -k "gate" 2>&1 | tail -5
  ;;
e)  # A/B: early touch of the next step's visited words (EHX_GW_WARM), same box, same index shape
  timeout 600 python -m pytest tests/test_graph_wide.py -x -q -k "models_walk" 2>&1 | tail -3
  graph_bench 6250k128_warm 6250000 128 l2 50,200,800 2,4
  EHX_LIB=$PWD/embeddinghub_amd/lib/libehx_nowarm.so graph_bench 6250k128_nowarm 6250000 128 l2 50,200,800 2,4
  graph_bench 6250k128_warm2 6250000 128 l2 50,200,800 2,4
  ;;
f)  # EHX-MANIFOLD-1 on the device == the oracle; then the default bench run with the 10 M x 768 structured graph leg
  timeout 600 python -m pytest tests/test_datagen.py -x -q 2>&1 | tail -3
  timeout 900 python bench.py > $O/r06_f_bench_line.json 2> $O/r06_f_bench.err; echo "bench rc=$?"
  cp $O/bench_detail.json $O/r06_f_bench_detail.json 2>/dev/null
  grep -E "leg done|structured|skipp" $O/r06_f_bench.err | tail -30
  python - <<PY
import json
l = json.load(open("$O/r06_f_bench_line.json"))
print(json.dumps({k: l.get(k) for k in ("value", "ms_per_step", "roofline", "graph_path", "graph_path_structured", "graph_path_structured_10m", "skipped")}, indent=0)[:3000])
d = json.load(open("$O/r06_f_bench_detail.json"))
for c in d.get("graph_path_structured_10m", {}).get("recall_vs_ef", []):
    print({k: c[k] for k in ("ef", "search_width", "recall_at_10", "value", "kernel_ms", "rows_fetched_per_query", "frac_of_8TBps")})
PY
  ;;
g)  # keyed rows after a generated base; configs[4] with the Set stream on the fp16 shard; lock-step A/B (EHX_I8_SYNC=2) on
    # the headline and under concurrent Sets, same box
  timeout 600 python -m pytest tests/test_generated_base.py tests/test_concurrent_set.py -x -q 2>&1 | tail -3
  for sync in 0 2 0 2; do
    EHX_I8_SYNC=$sync timeout 900 python bench.py --graph-rows 0 --structured-rows 0 --structured-big-rows 0 --single-query 0 \
      --reference-benchmark 0 --no-cpu-baseline --steps 20 --warmup 5 --detail-file $O/r06_g_sync${sync}_detail.json \
      > $O/r06_g_sync${sync}_line.json 2> $O/r06_g_sync${sync}.err; echo "sync=$sync rc=$?"
    python - <<PY
import json
l = json.load(open("$O/r06_g_sync${sync}_line.json"))
c4 = l["configs"]["configs[4]"]
print("sync=$sync headline %.3f ms/step kernel %.3f frac %.4f | one caller %s | c1 %.3f c3 %.3f c4 %.3f ms | set_concurrent(c4) %s | set_concurrent(1M) %s" % (
    l["ms_per_step"], l["roofline"]["kernel_ms"], l["roofline"]["frac"], l.get("one_caller_ms_per_step"),
    l["configs"]["configs[1]"]["ms_per_step"], l["configs"]["configs[3]"]["ms_per_step"], c4["ms_per_step"],
    c4.get("set_concurrent"), l.get("set_concurrent")))
PY
  done
  ;;
h)  # rocprofv3 evidence of round 6's code: the headline's launch trace + HBM traffic; the wide walk's trace and traffic on the
    # configs[3] shard and on the 10 M x 768 structured index
  R=$PWD
  mkdir -p $O/prof
  TAG=r06_h PASSES="trace fetch write" BENCH_ARGS="--config-legs 0" bash scripts/gpu_profile_i8.sh 2>&1 | tail -12
  cp $O/prof/r06_h_i8_*summary.txt $O/prof/r06_h_i8_traffic.json $O/ 2>/dev/null
  gtrace() {  # tag rows dims metric efs widths data rocprof-args...
    local tag=$1 rows=$2 dims=$3 metric=$4 efs=$5 widths=$6 data=$7; shift 7
    rm -rf $O/prof/$tag
    (cd /tmp && timeout 1500 rocprofv3 --kernel-trace "$@" -d $R/$O/prof/$tag -o p -- python $R/scripts/bench_graph.py --gpu-build \
       --rows $rows --dims $dims --metric $metric --efs $efs --widths $widths --data $data --reps 4 > $R/$O/r06_h_${tag}.jsonl 2> $R/$O/prof/$tag.err)
    python scripts/rocpd_summary.py $O/prof/$tag > $O/r06_h_${tag}_summary.txt 2>&1
    grep -E "graph_search|^kernel" $O/r06_h_${tag}_summary.txt | cut -c1-150 | head -8
  }
  gtrace graph_6250k128_trace 6250000 128 l2 200 1,4 gauss --stats
  gtrace graph_6250k128_fetch 6250000 128 l2 200 1,4 gauss --pmc FETCH_SIZE
  gtrace graph_s10m768_trace 10000000 768 cosine 60 1,4 manifold-dev:16 --stats
  gtrace graph_s10m768_fetch 10000000 768 cosine 60 1,4 manifold-dev:16 --pmc FETCH_SIZE
  find $O/prof -name "*.db" -size +8M -delete
  ;;
i)  # static issue priority for the second-resident wave of a SIMD (EHX_I8_PRIO build), same-box ABAB on three shapes; then
    # the whole GPU suite on the pruned library
  for rep in 1 2; do
    for lib in "" _prio; do
      for shape in "10000000 768 cosine" "6250000 128 l2" "1000000 768 cosine"; do
        set -- $shape
        EHX_LIB=$PWD/embeddinghub_amd/lib/libehx$lib.so timeout 600 python scripts/ab_flat.py --rows $1 --dims $2 --metric $3 --steps 40 --label "prio${lib:-_off}" 2>/dev/null >> $O/r06_i_prio_ab.jsonl
      done
    done
  done
  python - <<PY
import json
for l in open("$O/r06_i_prio_ab.jsonl"):
    r = json.loads(l)
    print("%-9s %9d x %4d  %.4f ms/step  kernel %.4f ms  checksum %d" % (r["label"], r["rows"], r["dims"], r["ms_per_step"], r["kernel_ms"], r["ids_checksum_last_batch"]))
PY
  timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
  ;;
j)  # wide walk, P = 4: ONE merge per step when at most 64 keys enter R
  timeout 1200 python -m pytest tests/test_graph_wide.py -x -q 2>&1 | tail -4
  graph_bench 6250k128 6250000 128 l2 50,200,800 1,2,4
  EHX_LIB=$PWD/embeddinghub_amd/lib/libehx_gprof.so graph_bench 6250k128_gprof 6250000 128 l2 200 4
  python - <<PY
import json
for l in open("$O/r06_j_graph_6250k128_gprof.jsonl"):
    r = json.loads(l)
    print("w=%d" % r["search_width"], "per step", r["phase_us_per_step"])
PY
  graph_bench 2m768 2000000 768 cosine 100,400 1,4
  ;;
k)  # wide walk: a helper wave per query for the row passes of short rows (on by shape; EHX_GRAPH_HELP=0: off), same box
  timeout 1200 python -m pytest tests/test_graph_wide.py -x -q -k "models_walk or contract" 2>&1 | tail -4
  EHX_GRAPH_HELP=0 graph_bench 6250k128_help0 6250000 128 l2 50,200,800 2,4
  graph_bench 6250k128_help1 6250000 128 l2 50,200,800 1,2,4
  EHX_GRAPH_HELP=0 graph_bench 6250k128_help0b 6250000 128 l2 200 4 --batches 2048
  graph_bench 6250k128_help1b 6250000 128 l2 200 4 --batches 2048
  EHX_GRAPH_HELP=1 graph_bench 2m768_help1 2000000 768 cosine 400 4
  graph_bench 2m768_help0 2000000 768 cosine 400 4
  ;;
l)  # the helper wave on the STRICT walk at short rows: parity with the oracle, then same-box A/B
  timeout 1500 python -m pytest tests/test_graph_parity.py tests/test_fuzz_graph.py tests/test_graph_wide.py -x -q 2>&1 | tail -4
  EHX_GRAPH_HELP=0 graph_bench 6250k128_help0 6250000 128 l2 50,200,800 1
  graph_bench 6250k128_help1 6250000 128 l2 50,200,800 1,4
  EHX_GRAPH_HELP=0 graph_bench 6250k128_help0r 6250000 128 l2 50,200,800 1
  ;;
m)  # what the HALF kernel's parked cycles are made of (6.25 M x 128: SQ counters, second set); then the default bench run of the
    # round's final library
  TAG=r06_m_6250k128 PASSES="sq sq2" BENCH_ARGS="--config-legs 0 --rows 6250000 --dims 128 --metric-kind l2 --steps 8" bash scripts/gpu_profile_i8.sh 2>&1 | tail -6
  grep -h "ScanArgsI8E" $O/prof/r06_m_6250k128_i8_pmc_sq_summary.txt $O/prof/r06_m_6250k128_i8_pmc_sq2_summary.txt | grep -v "^#" | cut -c1-170
  cp $O/prof/r06_m_*summary.txt $O/ 2>/dev/null
  find $O/prof -name "*.db" -size +4M -delete
  timeout 900 python bench.py > $O/r06_m_bench_line.json 2> $O/r06_m_bench.err; echo "bench rc=$?"
  cp $O/bench_detail.json $O/r06_m_bench_detail.json 2>/dev/null
  grep -E "leg done|structured|skipp" $O/r06_m_bench.err | tail -20
  ;;
n)  # the configs[3]-shard graph leg alone (6.25 M x 128 L2 structured rows: strict / wide walk with its helper wave), then the
    # new sharded test
  timeout 900 python bench.py --graph-rows 0 --structured-rows 0 --structured-big-rows 0 --single-query 0 --reference-benchmark 0 \
    --no-cpu-baseline --config-legs 0 --set-concurrent 0 --no-f32-engine --check-queries 0 --rows 1000000 \
    --detail-file $O/r06_n_detail.json > $O/r06_n_line.json 2> $O/r06_n.err; echo "bench rc=$?"
  python - <<PY
import json
d = json.load(open("$O/r06_n_detail.json"))
g = d["graph_path_structured_c3"]
print(g["workload"]); print(g["operating_point"], g["operating_point_strict_walk"], g["exact_flat_engine_same_rows_queries_per_s"], g["exact_truth_oracle_check"]["ids_identical_to_oracle"])
for c in g["recall_vs_ef"]:
    print({k: c[k] for k in ("ef", "search_width", "recall_at_10", "value", "kernel_ms", "rows_fetched_per_query", "frac_of_8TBps")})
PY
  timeout 900 python -m pytest tests/test_shards_abi.py -x -q 2>&1 | tail -3
  ;;
o)  # which exact engine serves structured rows fastest (the int8 filter on 128-dim structured L2 rows: 4.9 ms per batch)
  for shape in "6250000 128 l2 16" "6250000 128 l2 0" "6250000 128 cosine 16" "2000000 768 l2 16" "2000000 768 cosine 16"; do
    set -- $shape
    timeout 600 python scripts/structured_engine_ab.py --rows $1 --dims $2 --metric $3 --latent $4 2>/dev/null | tee -a $O/r06_o_structured_engine_ab.jsonl | cut -c1-230
  done
  ;;
p)  # int8 filter on rows whose norms vary (L2^2): group B margins + banded tile order — parity, then the engines on L2 shapes
  timeout 2400 python -m pytest tests/test_flat_parity.py tests/test_i8_filter.py tests/test_fuzz_parity.py tests/test_exactness.py tests/test_search_copy_layout.py tests/test_concurrent_set.py tests/test_generated_base.py -x -q 2>&1 | tail -4
  for shape in "6250000 128 l2 0" "6250000 128 l2 16" "2000000 768 l2 16" "6250000 128 cosine 0" "2000000 768 cosine 0"; do
    set -- $shape
    timeout 600 python scripts/structured_engine_ab.py --rows $1 --dims $2 --metric $3 --latent $4 --engines auto,f16 2>/dev/null | tee -a $O/r06_p_engine_ab.jsonl | cut -c1-200
  done
  ;;
q)  # launch-by-launch timeline of one batch of the int8 chain (device-resident queries, one stream): where a small shard's
    # and a short-row shard's batch goes.  SHAPES="rows dims metric;..." overrides.
  R=$PWD
  mkdir -p $O/prof
  IFS=';' read -ra SH <<< "${SHAPES:-1000000 768 cosine;1250000 768 cosine;6250000 128 l2;10000000 768 cosine}"
  for shape in "${SH[@]}"; do
    set -- $shape
    tag=r06_q_$1x$2_$3
    rm -rf $O/prof/$tag
    (cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $R/$O/prof/$tag -o p -- python $R/scripts/ab_flat.py --rows $1 --dims $2 --metric $3 $4 --steps ${QSTEPS:-24} --warmup 6 > $R/$O/prof/$tag.log 2>&1)
    tail -1 $O/prof/$tag.log | cut -c1-250
    python scripts/batch_timeline.py $O/prof/$tag | tee $O/${tag}_timeline.txt
  done
  find $O/prof -name "*.db" -size +8M -delete
  ;;
r)  # generic same-box ABAB of the exact flat path between library builds: LIBS="_prev :" (":" = the default library),
    # SHAPES="rows dims metric;...", REPS=2; PARITY=1 runs the flat parity suites on the default library first;
    # COUNT=1 prints the epilogue counters of the -DEHX_I8_COUNT build (libehx_count.so) per shape
  if [ "${PARITY:-1}" = 1 ]; then
    timeout 2400 python -m pytest tests/test_flat_parity.py tests/test_i8_filter.py tests/test_fuzz_parity.py tests/test_exactness.py tests/test_search_copy_layout.py tests/test_concurrent_set.py tests/test_generated_base.py tests/test_structured_rows.py -x -q 2>&1 | tail -4
  fi
  IFS=';' read -ra SH <<< "${SHAPES:-6250000 128 l2;10000000 768 cosine;1000000 768 cosine}"
  for rep in $(seq 1 ${REPS:-2}); do
    for lib in ${LIBS:-_prev :}; do
      [ "$lib" = ":" ] && lib=""
      for shape in "${SH[@]}"; do
        set -- $shape
        EHX_LIB=$PWD/embeddinghub_amd/lib/libehx$lib.so timeout 600 python scripts/ab_flat.py --rows $1 --dims $2 --metric $3 --steps 40 --label "${TAG:-ab}${lib:-_new}" 2>/dev/null >> $O/r06_r_${TAG:-ab}.jsonl
      done
    done
  done
  python - <<PY
import json
for l in open("$O/r06_r_${TAG:-ab}.jsonl"):
    r = json.loads(l)
    print("%-14s %9d x %4d %-6s %.4f ms/step  kernel %.4f ms  checksum %d  fallbacks %d/%d" % (r["label"], r["rows"], r["dims"], r["metric"], r["ms_per_step"], r["kernel_ms"], r["ids_checksum_last_batch"], r["i8_fallback"], r["filter_fallback"]))
PY
  if [ "${COUNT:-0}" = 1 ]; then
    for shape in "${SH[@]}"; do
      set -- $shape
      echo "== epilogue counters, $shape"
      EHX_LIB=$PWD/embeddinghub_amd/lib/libehx_count.so EHX_I8_COUNT=1 timeout 600 python scripts/ab_flat.py --rows $1 --dims $2 --metric $3 --steps 4 --warmup 2 2>&1 | grep -a "i8 count" | tail -2 | cut -c1-200
    done
  fi
  ;;
u)  # the round's final library: whole GPU suite, rocprofv3 trace + FETCH / WRITE of the headline, trace + SQ counters of the
    # configs[3] shard shape, batch timelines of four shapes, the default bench run
  timeout 2700 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $O/r06_u_pytest_gpu_tail.txt
  TAG=r06_u PASSES="trace fetch write" BENCH_ARGS="--config-legs 0 --structured-c3-rows 0" bash scripts/gpu_profile_i8.sh 2>&1 | tail -12
  cp $O/prof/r06_u_i8_*summary.txt $O/prof/r06_u_i8_traffic.json $O/ 2>/dev/null
  TAG=r06_u_6250k128 PASSES="trace sq" BENCH_ARGS="--config-legs 0 --structured-c3-rows 0 --rows 6250000 --dims 128 --metric-kind l2 --steps 8" bash scripts/gpu_profile_i8.sh 2>&1 | tail -4
  cp $O/prof/r06_u_6250k128_*summary.txt $O/ 2>/dev/null
  find $O/prof -name "*.db" -size +4M -delete
  S=q bash scripts/gpu_r06.sh q 2>&1 | grep -v "^W2026" | grep "total\|^#"
  for f in $O/r06_q_*_timeline.txt; do cp $f $O/r06_u_$(basename $f | sed s/r06_q_//); done
  timeout 1200 python bench.py > $O/r06_u_bench_line.json 2> $O/r06_u_bench.err; echo "bench rc=$?"
  cp $O/bench_detail.json $O/r06_u_bench_detail.json 2>/dev/null
  grep -E "leg done|skipp" $O/r06_u_bench.err | tail -20 | cut -c1-220
  python - <<PY
import json
l = json.load(open("$O/r06_u_bench_line.json"))
print(json.dumps({k: l.get(k) for k in ("value", "ms_per_step", "roofline", "one_caller_ms_per_step", "device_resident_ms_per_step")})[:900])
for k, v in l.get("configs", {}).items():
    print(k, {kk: v.get(kk) for kk in ("ms_per_step", "frac", "kernel_ms", "oracle")})
PY
  ;;
v)  # the driver's own sequence on the final library: smoke(), then bench.py with the driver's arguments; and the SQ counters
    # of the headline's main pass (the wave-cycle split DESIGN quotes)
  timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
  timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r06_${VT:-v}_bench_line.json 2> $O/r06_${VT:-v}_bench.err; echo "bench rc=$?"
  cp $O/bench_detail.json $O/r06_${VT:-v}_bench_detail.json 2>/dev/null
  python - <<PY
import json
l = json.load(open("$O/r06_${VT:-v}_bench_line.json"))
print(json.dumps({k: l.get(k) for k in ("value", "ms_per_step", "steps", "warmup", "roofline", "one_caller_ms_per_step", "device_resident_ms_per_step")})[:900])
for k, v in l.get("configs", {}).items():
    print(k, {kk: v.get(kk) for kk in ("ms_per_step", "frac", "kernel_ms", "oracle")})
print("line bytes", len(open("$O/r06_${VT:-v}_bench_line.json").read()))
PY
  TAG=r06_${VT:-v} PASSES="sq" BENCH_ARGS="--config-legs 0 --structured-c3-rows 0" bash scripts/gpu_profile_i8.sh 2>&1 | tail -3
  grep -v "^#   " $O/prof/r06_${VT:-v}_i8_pmc_sq_summary.txt > $O/r06_${VT:-v}_i8_pmc_sq_summary.txt
  grep "Lb0ELb1ELb0ELb0E" $O/r06_${VT:-v}_i8_pmc_sq_summary.txt | cut -c1-150
  find $O/prof -name "*.db" -size +4M -delete
  ;;
*) echo "unknown session $S"; exit 2;;
esac
