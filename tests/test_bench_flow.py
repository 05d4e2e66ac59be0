"""Dry run of bench.py's main() on the CPU (no GPU, no engine): the engine, torch.cuda and the sharded searcher are
replaced by small numpy / oracle stand-ins so that the WHOLE control flow runs — timed region, exactness block (ten
batches against the 'fp32 engine', the oracle check), CPU baseline, the optional legs under their time budget and
watchdog, the one JSON line.  What this pins is bench.py itself (names, order, the JSON contract of the driver), not
the engine: the stand-in answers every kNN with the oracle's exhaustive scan."""
import importlib.util
import io
import json
import os
import sys
from contextlib import redirect_stdout

import numpy as np
import pytest
import torch

from oracle import pyoracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


sys.path.insert(0, os.path.join(ROOT, "tests", "bench_stubs"))
import standins  # noqa: E402  (tests/bench_stubs/standins.py: the engine / torch.cuda stand-ins)
from standins import FakeSpace  # noqa: E402


@pytest.fixture
def bench_on_stand_ins(monkeypatch, tmp_path):
    standins.install(monkeypatch)
    spec = importlib.util.spec_from_file_location("bench_flow_module", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for var in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(var, raising=False)

    def run(argv):
        """runs main(); returns the FULL record (the detail file).  The one stdout line — the compact record the driver
        parses — is checked here for every run (size bound, contract fields) and kept as run.line"""
        detail = str(tmp_path / "bench_detail.json")
        monkeypatch.setattr(sys, "argv", ["bench.py"] + argv + ["--detail-file", detail])
        buf = io.StringIO()
        with redirect_stdout(buf):
            bench.main()
        lines = [l for l in buf.getvalue().splitlines() if l.strip()]
        assert len(lines) == 1, "bench.py must print exactly ONE line on stdout: %r" % lines
        assert len(lines[0].encode()) <= 4096, "the stdout line is bounded (VERDICT r04 #1): %d bytes" % len(lines[0])
        run.line = json.loads(lines[0])
        assert run.line["detail"] == detail
        with open(detail) as f:
            return json.load(f)
    run.module = bench
    run.line = None
    return run


SMALL = ["--structured-big-rows", "0", "--structured-c3-rows", "0", "--rows", "3000", "--dims", "32", "--batch", "16", "--steps", "3", "--warmup", "1", "--check-queries", "8",
         "--cpu-sample-rows", "3000", "--cpu-sample-queries", "8", "--cpu-hnsw-rows", "500", "--graph-batches", "2",
         "--graph-efs", "10,20", "--reference-benchmark", "0", "--config-legs", "0", "--single-query", "0",
         "--set-concurrent", "0"]


def test_default_flow_prints_one_json_line_with_the_contract_fields(bench_on_stand_ins):
    r = bench_on_stand_ins(SMALL + ["--graph-rows", "1000", "--structured-rows", "600"])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in r, key
    assert r["n_gpus"] == 1 and r["steps"] == 3 and r["warmup"] == 1 and r["higher_is_better"] is True
    assert r["vs_baseline"] is None and r["data"] == "synthetic" and "workload" in r["config"]
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(r["roofline"])
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(r["cpu_baseline"]) and r["cpu_baseline"]["kind"] == "port"
    ex = r["exactness"]
    assert ex["filter_vs_f32_engine_identical"] is True and ex["filter_vs_f32_engine_queries"] == 4 * 16  # all 4 resident batches
    assert ex["recall_at_10"] == 1.0 and ex["ids_identical_to_oracle"] and ex["dist_bytes_identical_to_oracle"]
    assert ex["oracle_rows"] == 3000 and "oracle_partial" not in ex
    assert r["graph_path"]["recall_vs_ef"][0]["ef"] == 10 and "operating_point" in r["graph_path"]
    assert "graph_path_structured" in r and r["optional_legs_skipped"] == {}
    cs = r["graph_path_structured"]["cpu_hnsw_sample_in_run"]   # the CPU oracle's HNSW over the same rows, built in the run
    assert cs["rows"] == 600 and [p["ef"] for p in cs["recall_vs_ef"]] == [10, 20, 40, 60, 100, 200] and cs["threads"] >= 1
    assert all(p["qps_all_cores"] > 0 and 0.0 <= p["recall_at_10"] <= 1.0 for p in cs["recall_vs_ef"])
    assert "device_resident_queries" in r and "host_pointer_one_caller" in r and "f32_scan_engine" in r
    assert "host pointers" in r["config"]["timed_from"] and r["exactness"]["host_pointer_path_identical"] is True
    keys = list(r)                       # quoted tables last, the measured legs before them
    assert keys[-1] == "cpu_hnsw_offline" and keys.index("roofline") < keys.index("cpu_baseline")
    # ---- the compact stdout line: the contract + one number set per leg, nothing else ----
    c = bench_on_stand_ins.line
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "exactness"):
        assert key in c, key
        if key not in ("config", "roofline", "cpu_baseline", "exactness"):
            assert c[key] == r[key], key
    assert set(("workload", "timed_from", "rows", "dims", "batch", "k", "world_size")) <= set(c["config"])
    assert set(("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_session", "kernel_ms")) <= set(c["roofline"])
    assert c["roofline"]["frac"] == r["roofline"]["frac"] and c["roofline"]["kernel"] == "flat_scan_i8_kernel"
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(c["cpu_baseline"]) and c["cpu_baseline"]["kind"] == "port"
    assert len(c["cpu_baseline"]["sample"]) < 160 and c["cpu_baseline"]["value"] == r["cpu_baseline"]["value"]
    assert c["exactness"]["ids_identical_to_oracle"] and c["exactness"]["dist_bytes_identical_to_oracle"]
    assert c["exactness"]["recall_at_10"] == 1.0 and c["recall_at_10"] == 1.0
    for leg in ("graph_path", "graph_path_structured"):
        assert set(("ef", "recall_at_10", "qps", "frac")) <= set(c[leg]) and "recall_vs_ef" not in c[leg]
    assert c["graph_path_structured"]["exact_flat_qps"] == r["graph_path_structured"]["exact_flat_engine_same_rows_queries_per_s"]
    assert not any(isinstance(v, str) and len(v) > 200 for v in json.dumps(c).split('"'))   # no prose


def test_structured_10m_leg_reports_the_graph_path_at_the_metrics_recall_with_both_walks(bench_on_stand_ins):
    """BASELINE configs[2] on rows a graph can index: EHX-MANIFOLD-1 rows from the generator (no host
    array in the real run), exact truth checked against the oracle (fail-closed), ef swept for the strict and the wide
    walk, the operating point and both walks' numbers in the compact line"""
    r = bench_on_stand_ins(SMALL[4:] + ["--structured-big-rows", "900", "--structured-big-efs", "10,20", "--graph-rows", "0",
                                        "--structured-rows", "0", "--no-cpu-baseline", "--structured-c3-rows", "700"])
    g = r["graph_path_structured_10m"]
    assert g["rows"] == 900 and g["exact_truth_identical_to_oracle"] is True
    assert g["exact_truth_oracle_check"]["oracle_rows"] == 900
    assert {(c["ef"], c["search_width"]) for c in g["recall_vs_ef"]} == {(10, 1), (10, 4), (20, 1), (20, 4)}
    assert g["operating_point"]["meets_recall_0.95"] is True      # (the stand-in answers exhaustively)
    c = bench_on_stand_ins.line["graph_path_structured_10m"]
    assert c["oracle"] is True and c["meets_recall_0.95"] is True and c["rows"] == 900
    assert "strict_qps" in c and "width" in c and "exact_flat_qps" in c
    g3 = r["graph_path_structured_c3"]     # the configs[3] shard shape: 128-dim L2 rows, not normalised
    assert g3["rows"] == 700 and g3["exact_truth_identical_to_oracle"] is True and "x128 l2" in g3["workload"]
    assert bench_on_stand_ins.line["graph_path_structured_c3"]["meets_recall_0.95"] is True


def test_compact_line_stays_bounded_when_every_leg_reports(bench_on_stand_ins):
    """worst case for the line's size: every optional leg on (config legs, single query, set concurrent, set stream,
    both graph legs): still one line of <= 4096 bytes with the per-leg summaries (asserted inside run())"""
    argv = [a for a in SMALL]
    for flag, v in (("--config-legs", "1"), ("--single-query", "6"), ("--set-concurrent", "64")):
        argv[argv.index(flag) + 1] = v
    r = bench_on_stand_ins(argv + ["--config-legs-rows-div", "5000", "--graph-rows", "1000", "--structured-rows", "600",
                                   "--set-stream", "64"])
    c = bench_on_stand_ins.line
    assert "truncated" not in c
    assert list(c["configs"]) == ["configs[1]", "configs[3]", "configs[4]"]
    for name, leg in c["configs"].items():
        assert leg["oracle"] is True and leg["ms_per_step"] == r["configs"][name]["ms_per_step"]
        assert leg["frac"] == r["configs"][name]["roofline"]["frac"]
    assert set(("flat_us", "graph_ef10_us", "cpu_hnsw_us")) <= set(c["single_query"])
    assert set(("set_rows_per_s", "search_fraction_of_alone")) <= set(c["set_concurrent"])
    # and the fallback: an absurdly long value in a leg summary drops optional legs, never the contract
    bench = bench_on_stand_ins.module
    big = dict(r)
    big["configs"] = {("configs[%d]" % i): dict(r["configs"]["configs[1]"]) for i in range(200)}
    s = bench.compact_line(big)
    assert len(s) <= 4096
    back = json.loads(s)
    assert back["truncated"] is True and "roofline" in back and "cpu_baseline" in back and "configs" not in back


def test_config_legs_report_three_more_shapes_with_exactness_and_roofline_before_the_cpu_tables(bench_on_stand_ins):
    """configs[1], the configs[3] shard and the configs[4] shard (fp16 rows: the oracle must scan the
    ROUNDED rows) are default legs, each with `exactness` (ids and distance bytes identical) and `roofline`, placed before
    the CPU-baseline tables"""
    monkey_argv = sys.argv
    sys.argv = ["bench.py"]
    try:
        assert bench_on_stand_ins.module.parse().config_legs == 1      # the default itself
    finally:
        sys.argv = monkey_argv
    argv = [a for a in SMALL]
    argv[argv.index("--config-legs") + 1] = "1"
    r = bench_on_stand_ins(argv + ["--config-legs-rows-div", "5000", "--graph-rows", "0", "--structured-rows", "0"])
    legs = r["configs"]
    assert list(legs) == ["configs[1]", "configs[3]", "configs[4]"]
    for name, rows, dims in (("configs[1]", 200, 768), ("configs[3]", 1250, 128), ("configs[4]", 2500, 1536)):
        leg = legs[name]
        assert leg["workload"].startswith("%dx%d " % (rows, dims)) and leg["value"] > 0 and "host pointers" in leg["timed_from"]
        assert set(("bound", "achieved", "peak", "unit", "frac", "kernel_ms")) <= set(leg["roofline"])
        ex = leg["exactness"]
        assert ex["ids_identical_to_oracle"] is True and ex["dist_bytes_identical_to_oracle"] is True
        assert ex["oracle_rows"] == rows and ex["recall_at_10"] == 1.0
    assert "rows f16" in legs["configs[4]"]["workload"]
    keys = list(r)
    assert keys.index("configs") < keys.index("cpu_baseline")


def test_configs4_leg_streams_sets_into_the_fp16_shard_while_searching_and_checks_both_sides(bench_on_stand_ins):
    """VERDICT r05 #3 — BASELINE configs[4] as written, "streamed Set + concurrent GetNeighbors" on the fp16 x 1536 shard:
    four writer threads stream chunks through set_batch into the SAME space the callers search; a batch answered during
    the writes is checked against the oracle's truth over the rows that were complete, one answered after the last Set
    against its truth over everything; rows/s and the search rate beside the un-contended one are in the line"""
    argv = [a for a in SMALL]
    argv[argv.index("--config-legs") + 1] = "1"
    argv[argv.index("--set-concurrent") + 1] = "131072"
    r = bench_on_stand_ins(argv + ["--config-legs-rows-div", "5000", "--graph-rows", "0", "--structured-rows", "0",
                                   "--no-cpu-baseline"])
    sc = r["configs"]["configs[4]"]["set_concurrent"]
    assert sc["rows_written_meanwhile"] == 131072 // 5000 and sc["writer_threads"] == 4 and sc["error"] is None
    assert sc["oracle_during_writes"]["consistent_with_the_complete_prefix"] is True
    assert sc["oracle_after_writes"]["ok"] is True and "identical" in sc["oracle_after_writes"]["how"]
    assert sc["set_rows_per_s_meanwhile"] > 0 and sc["search_batches_meanwhile"] >= 2
    c = bench_on_stand_ins.line["configs"]["configs[4]"]["set_concurrent"]
    assert c["oracle_during"] is True and c["oracle_after"] is True and c["writers"] == 4
    assert "set_concurrent" not in r["configs"]["configs[1]"]


def test_an_oracle_mismatch_ends_the_run(bench_on_stand_ins, monkeypatch):
    """fail-closed: ids or distance bytes that differ from the oracle's abort the run — recall alone
    does not decide.  Here the fp16 leg's stand-in space 'forgets' to round its rows: same ids almost everywhere, other
    distance bytes."""
    monkeypatch.setattr(FakeSpace, "half", False, raising=False)
    real_init = FakeSpace.__init__

    def init(self, *a, **kw):
        real_init(self, *a, **kw)
        self.half = False
    monkeypatch.setattr(FakeSpace, "__init__", init)
    argv = [a for a in SMALL]
    argv[argv.index("--config-legs") + 1] = "1"
    with pytest.raises(SystemExit) as e:
        bench_on_stand_ins(argv + ["--config-legs-rows-div", "5000", "--graph-rows", "0", "--structured-rows", "0",
                                   "--no-cpu-baseline"])
    assert "EXACTNESS FAILURE" in str(e.value) and "f16 rows" in str(e.value)


def test_set_concurrent_is_a_default_leg(bench_on_stand_ins):
    """streamed Set concurrent with search (BASELINE configs[4]) must be driver-visible: on by default"""
    monkey_argv = sys.argv
    sys.argv = ["bench.py"]
    try:
        assert bench_on_stand_ins.module.parse().set_concurrent > 0      # the default itself
    finally:
        sys.argv = monkey_argv
    r = bench_on_stand_ins(SMALL[:-2] + ["--set-concurrent", "4096", "--graph-rows", "0", "--structured-rows", "0",
                                         "--no-cpu-baseline"])
    sc = r["set_concurrent"]
    assert sc["rows_written_meanwhile"] == 4096 and sc["error"] is None
    assert sc["search_batches_meanwhile"] >= 1 and sc["set_rows_per_s_meanwhile"] > 0
    assert "set_concurrent" not in r["optional_legs_skipped"]


def test_single_query_leg_reports_configs0_and_checks_both_paths_against_the_oracle(bench_on_stand_ins):
    """BASELINE configs[0] (10 k x 128 L2, one query per call) is a default leg: latency of the exact path and of graph
    mode at the reference's ef = 10, each compared with the oracle before it is reported"""
    monkey_argv = sys.argv
    sys.argv = ["bench.py"]
    try:
        assert bench_on_stand_ins.module.parse().single_query > 0
    finally:
        sys.argv = monkey_argv
    r = bench_on_stand_ins(SMALL[:-4] + ["--single-query", "12", "--set-concurrent", "0", "--graph-rows", "0",
                                         "--structured-rows", "0", "--no-cpu-baseline"])
    sq = r["single_query"]
    assert sq["calls"] == 12 and "10000 x 128 L2" in sq["workload"]
    assert sq["flat_exact"]["identical_to_oracle_exhaustive"] == "12 of 12" and sq["flat_exact"]["latency_us_median"] > 0
    assert sq["graph_ef10"]["identical_to_oracle_hnsw"] == "12 of 12" and sq["graph_ef10"]["cpu_oracle_hnsw_us_1_thread"] > 0
    assert "single_query" not in r["optional_legs_skipped"]


def test_a_spent_time_budget_skips_the_single_query_leg(bench_on_stand_ins):
    r = bench_on_stand_ins(SMALL[:-4] + ["--single-query", "12", "--set-concurrent", "0", "--graph-rows", "0",
                                         "--structured-rows", "0", "--no-cpu-baseline", "--time-budget", "0"])
    assert "single_query" not in r and "single_query" in r["optional_legs_skipped"]


def test_a_spent_time_budget_skips_the_optional_legs_and_says_so(bench_on_stand_ins):
    r = bench_on_stand_ins(SMALL + ["--graph-rows", "1000", "--structured-rows", "600", "--time-budget", "0"])
    assert "graph_path" not in r and "graph_path_structured" not in r
    assert set(r["optional_legs_skipped"]) == {"graph_path", "graph_path_structured"}
    assert "cpu_baseline" in r and r["exactness"]["recall_at_10"] == 1.0   # the required legs ran


def test_a_spent_time_budget_skips_the_set_concurrent_leg_too(bench_on_stand_ins):
    r = bench_on_stand_ins(SMALL[:-2] + ["--set-concurrent", "4096", "--graph-rows", "0", "--structured-rows", "0",
                                         "--time-budget", "0", "--no-cpu-baseline"])
    assert "set_concurrent" not in r and "set_concurrent" in r["optional_legs_skipped"]


def test_a_slow_host_gets_the_prefix_check(bench_on_stand_ins, monkeypatch):
    # an oracle that can only afford its first chunk: the run must still finish, flagged as partial, still consistent
    bench = bench_on_stand_ins.module
    real = bench.oracle_truth
    monkeypatch.setattr(bench, "oracle_truth", lambda *a, **kw: real(*a, **dict(kw, max_seconds=0.0)))
    r = bench_on_stand_ins(SMALL + ["--rows", "1200000", "--dims", "8", "--graph-rows", "0", "--structured-rows", "0",
                                    "--no-cpu-baseline", "--check-queries", "4"])
    ex = r["exactness"]
    assert ex.get("oracle_partial") is True and 0 < ex["oracle_rows"] < 1200000
    assert ex["ids_identical_to_oracle"] is True and ex["recall_at_10"] == 1.0


def test_a_stuck_optional_leg_is_abandoned_by_the_watchdog(bench_on_stand_ins, monkeypatch):
    import threading
    import time
    bench = bench_on_stand_ins.module
    exits = []

    class QuickTimer(threading.Timer):          # the watchdog's interval shrunk to 0.2 s for the test
        def __init__(self, interval, function, *a, **kw):
            super().__init__(0.2, function, *a, **kw)
    monkeypatch.setattr(bench.threading, "Timer", QuickTimer)
    monkeypatch.setattr(bench.os, "_exit", lambda code: exits.append(code))   # (the real one ends the process)
    real_fill = FakeSpace.fill_synthetic

    def slow_fill(self, seed, row0, n, normalize):
        if self.mode == 1:                      # the graph leg's index: "stuck"
            time.sleep(0.8)
        real_fill(self, seed, row0, n, normalize)
    monkeypatch.setattr(FakeSpace, "fill_synthetic", slow_fill)
    r = bench_on_stand_ins(SMALL + ["--graph-rows", "1000", "--structured-rows", "0", "--no-cpu-baseline"])
    assert exits == [0]
    assert "graph_path" not in r and "watchdog" in r["optional_legs_skipped"]["graph_path"]
    assert r["exactness"]["recall_at_10"] == 1.0 and r["value"] > 0       # the headline part is all there


def test_multi_rank_flow_rank0_prints_the_line(bench_on_stand_ins, monkeypatch):
    """the path torchrun drives (WORLD_SIZE > 1): barriers, MAX-reduced time, no single-GPU legs — collectives stubbed"""
    import torch.distributed as dist
    calls = []
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("LOCAL_RANK", "0")
    monkeypatch.setattr(dist, "init_process_group", lambda *a, **kw: calls.append("init"))
    monkeypatch.setattr(dist, "barrier", lambda *a, **kw: calls.append("barrier"))
    monkeypatch.setattr(dist, "all_reduce", lambda t, op=None: calls.append("all_reduce"))
    monkeypatch.setattr(dist, "all_gather", lambda lst, t: [x.copy_(t) for x in lst] and calls.append("all_gather"))
    monkeypatch.setattr(dist, "destroy_process_group", lambda *a, **kw: calls.append("destroy"))
    monkeypatch.setattr(dist, "get_world_size", lambda *a, **kw: 2)
    real_tensor, real_device = torch.tensor, torch.device
    monkeypatch.setattr(torch, "tensor", lambda *a, **kw: real_tensor(*a, **{k: v for k, v in kw.items() if k != "device"}))
    monkeypatch.setattr(torch, "device", lambda *a, **kw: real_device("cpu"))
    # the stand-in shard holds every row (shard_range stub), so the merged answer is the exact one
    r = bench_on_stand_ins(SMALL + ["--gpus", "2"])
    assert r["n_gpus"] == 2 and "row-shard x2" in r["config"]["parallelism"]
    assert "graph_path" not in r and "cpu_baseline" not in r          # N = 1 legs
    assert r["exactness"]["recall_at_10"] == 1.0
    assert calls[0] == "init" and calls[-1] == "destroy" and "all_reduce" in calls and calls.count("barrier") >= 3


def _run_bench_subprocess(argv, devices, detail=None):
    import subprocess
    if detail is not None:
        argv = argv + ["--detail-file", str(detail)]
    env = dict(os.environ, EHX_BENCH_STANDINS="1", EHX_STANDIN_DEVICES=str(devices),
               PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "tests", "bench_stubs"), ROOT,
                                           os.environ.get("PYTHONPATH", "")]))
    for var in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(var, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=env, capture_output=True,
                          text=True, timeout=600)


def test_gpus_2_launches_two_real_ranks(tmp_path):
    """`python bench.py --gpus 2` (no torchrun around it, WORLD_SIZE unset) must start the two ranks itself.
    Here: the real launcher, two real processes under torch.distributed.run, a real process group (gloo instead of RCCL),
    the product's own sharded.py (row partition, ONE packed all-gather per batch, merge) — only the engine and torch.cuda
    are stand-ins (sitecustomize.py in tests/bench_stubs).  Each rank holds half of the rows; the merged answer is checked
    against the oracle's exhaustive scan of ALL rows inside bench.py (exactness block), so a wrong partition, gather or
    merge fails the run."""
    r = _run_bench_subprocess(SMALL + ["--gpus", "2", "--rows", "3001"], devices=2, detail=tmp_path / "d.json")
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    # the JSON line is the LAST line on stdout even though the ranks' collective library writes there through C stdio
    # (RCCL's "Librccl path : ..." sits in a block buffer until exit): rank 0 flushes it first, the others write to stderr
    assert r.stdout.strip().splitlines()[-1] == lines[0], r.stdout[-500:]
    assert "Librccl path : stand-in (rank 0)" in r.stdout + r.stderr
    line = json.loads(lines[0])
    assert len(lines[0]) <= 4096 and line["n_gpus"] == 2 and line["config"]["world_size"] == 2
    assert line["config"]["rows"] == 3001 and "row-shard x2" in line["config"]["parallelism"] and line["scaling"] == "strong"
    assert line["exactness"]["ids_identical_to_oracle"] and line["exactness"]["recall_at_10"] == 1.0
    assert "roofline" in line and line["value"] > 0
    with open(tmp_path / "d.json") as f:
        out = json.load(f)
    assert out["n_gpus"] == 2 and out["config"]["world_size"] == 2
    assert out["config"]["rows_per_rank"] == [1500, 1501] and out["config"]["rows_per_gpu"] == 1500
    assert out["scaling"] == "strong" and "row-shard x2" in out["config"]["parallelism"]
    ex = out["exactness"]
    assert ex["recall_at_10"] == 1.0 and ex["ids_identical_to_oracle"] and ex["dist_bytes_identical_to_oracle"]
    assert ex["oracle_rows"] == 3001
    assert "launching 2 ranks" in r.stderr


def test_gpus_4_dry_run_ranks_partition_gather_and_only_rank0_does_host_legs(tmp_path):
    """the shape of the driver's scaling run, at 4 ranks — real launcher, four real processes, gloo
    collectives, the product's sharded.py.  The shards are uneven (3003 rows over 4 ranks), every rank is handed its
    batches as HOST buffers (double-buffered upload path), the merged answer is checked against the oracle over all
    rows, rank 0 alone prints (and runs no single-GPU legs: no cpu_baseline, no graph legs, no config legs), and the
    line reports the world size the process group itself returns."""
    r = _run_bench_subprocess(SMALL + ["--gpus", "4", "--rows", "3003"], devices=4, detail=tmp_path / "d.json")
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert len(lines[0]) <= 4096 and line["n_gpus"] == 4 and line["config"]["world_size"] == 4
    for leg in ("cpu_baseline", "graph_path", "graph_path_structured", "configs", "single_query", "set_concurrent"):
        assert leg not in line, leg
    with open(tmp_path / "d.json") as f:
        out = json.load(f)
    assert out["n_gpus"] == 4 and out["config"]["world_size"] == 4 and out["scaling"] == "strong"
    assert out["config"]["rows_per_rank"] == [750, 750, 750, 753] and sum(out["config"]["rows_per_rank"]) == 3003
    assert "pinned host tensors" in out["config"]["timed_from"] and "row-shard x4" in out["config"]["parallelism"]
    ex = out["exactness"]
    assert ex["recall_at_10"] == 1.0 and ex["ids_identical_to_oracle"] and ex["dist_bytes_identical_to_oracle"]
    assert ex["oracle_rows"] == 3003
    for leg in ("cpu_baseline", "graph_path", "graph_path_structured", "configs", "single_query", "set_concurrent"):
        assert leg not in out, leg
    assert "device_resident_queries" in out        # (the same steps from HBM-resident batches, every rank)


def test_gpus_8_dry_run_the_drivers_scaling_shape(tmp_path):
    """`bench.py --gpus 8` as the driver's SCALE run launches it, dry — eight real processes under
    torch.distributed.run, gloo collectives, the product's sharded.py; uneven shards (3005 rows over 8 ranks), rank 0 alone
    on the host legs, the world size the process group reports, the rows of all ranks adding up, and every rank's own
    ms_per_step in the line so that a straggler is visible in the first real run.  (No hardware claim: SCALE stays
    "unmeasured" until a record with N > 1 exists.)"""
    import time as _t
    t0 = _t.time()
    r = _run_bench_subprocess(SMALL + ["--gpus", "8", "--rows", "3005"], devices=8, detail=tmp_path / "d.json")
    wall = _t.time() - t0
    assert r.returncode == 0, r.stderr[-3000:]
    assert wall < 300, wall
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert len(lines[0]) <= 4096 and line["n_gpus"] == 8 and line["config"]["world_size"] == 8
    assert len(line["config"]["rank_ms"]) == 8 and all(x > 0 for x in line["config"]["rank_ms"])
    assert line["config"]["rank_spread"] >= 1.0
    assert line["exactness"]["ids_identical_to_oracle"] and line["exactness"]["recall_at_10"] == 1.0
    for leg in ("cpu_baseline", "graph_path", "graph_path_structured", "graph_path_structured_10m", "configs", "single_query",
                "set_concurrent"):
        assert leg not in line, leg
    with open(tmp_path / "d.json") as f:
        out = json.load(f)
    assert out["config"]["world_size"] == 8 and sum(out["config"]["rows_per_rank"]) == 3005 == out["config"]["rows_total"]
    assert len(set(out["config"]["rows_per_rank"])) > 1          # uneven shards
    assert out["scaling"] == "strong" and "row-shard x8" in out["config"]["parallelism"]
    assert "launching 8 ranks" in r.stderr


def test_gpus_2_on_a_one_gpu_box_fails_loudly():
    r = _run_bench_subprocess(SMALL + ["--gpus", "2"], devices=1)
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert "--gpus 2" in r.stderr and "1 GPU(s) visible" in r.stderr


def test_world_size_that_contradicts_gpus_is_an_error(bench_on_stand_ins, monkeypatch):
    monkeypatch.setenv("WORLD_SIZE", "4")
    with pytest.raises(SystemExit) as e:
        bench_on_stand_ins(SMALL + ["--gpus", "2"])
    assert "WORLD_SIZE=4" in str(e.value)


def test_reference_benchmark_leg_over_two_oracle_stores():
    """the reference's benchmark shape (benchmark.py:217-272) through the gRPC shim — MultiSet, sequential
    nearest_neighbor(num=20, key=...) RPCs, the same through a 10-worker pool, two stores side by side, key lists
    compared.  Here (no GPU) both sides are the oracle-backed store: pins the leg's flow and its record."""
    import importlib.util
    from oracle.oracle_store import OracleStore
    spec = importlib.util.spec_from_file_location("bench_refbench_module", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    r = bench.reference_benchmark(OracleStore, OracleStore, rows=400, dims=50, requests=120, num=20, workers=10)
    assert r["identical_key_lists"] is True and r["lists_compared"] == 240 and "first_difference" not in r
    for k in ("engine_multiset_s", "engine_sequential_s", "engine_pool10_s", "oracle_sequential_s", "oracle_pool10_s"):
        assert r[k] > 0, k
    line = bench.compact({"reference_benchmark": r})
    assert line["reference_benchmark"]["identical_key_lists"] is True and line["reference_benchmark"]["requests"] == 120
    assert set(line["reference_benchmark"]) == {"rows", "dims", "requests", "engine_sequential_s", "engine_pool10_s",
                                                "oracle_sequential_s", "oracle_pool10_s", "identical_key_lists"}
