cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 1200 python -m pytest tests/test_concurrent_set.py tests/test_generated_base.py tests/test_flat_parity.py -x -q 2>&1 | tail -2
for rep in 1 2; do
  timeout 900 python bench.py --graph-rows 0 --structured-rows 0 --structured-big-rows 0 --structured-c3-rows 0 --single-query 0 \
      --reference-benchmark 0 --no-cpu-baseline --no-f32-engine --rows 1000000 --detail-file $O/_y_$rep.json > $O/_y_${rep}_line.json 2> $O/_y_$rep.err
  python - <<PY
import json
d = json.load(open("$O/_y_$rep.json"))
c4 = d["configs"]["configs[4]"]
sc = c4.get("set_concurrent", {})
print("configs[4] %.3f ms/step  set_concurrent: %s" % (c4["ms_per_step"], json.dumps(sc)[:700]))
s1 = d.get("set_concurrent", {})
print("         1M x 768 set_concurrent: %s" % json.dumps({k: s1.get(k) for k in ("set_rows_per_s_meanwhile", "search_fraction_of_alone", "error")}))
PY
done
