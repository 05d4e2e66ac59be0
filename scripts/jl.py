#!/usr/bin/env python3
"""Compact view of bench_graph.py JSON lines (one row per line)."""
import json
import sys
for line in open(sys.argv[1]):
    try:
        r = json.loads(line)
    except ValueError:
        continue
    print(r["workload"].split(",")[0], r["workload"].split("ef=")[-1], "kernel_ms", r["kernel_ms"], "qps", r["qps_kernel"],
          "recall", r["recall_at_k"], "ndist", r["n_dist_per_query"], "hops", r["n_hops_per_query"],
          "GB/s", r["roofline"]["achieved"], "frac", r["roofline"]["frac"], "pf_hit", r.get("prefetch_hit_rate"),
          "same_as_oracle", r.get("queries_identical_to_oracle"), "phase_us", r.get("phase_us_per_hop"), "flat_qps", r.get("flat_exact_qps_host_pointers"))
