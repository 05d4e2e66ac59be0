#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite result (kernel-trace / --pmc runs) as plain text.

usage: rocpd_summary.py results.db [more.db ...] > profiles/summary.txt
Per kernel: calls, total/avg/min/max duration (from the kernel dispatch records), registers and
LDS as recorded; per (kernel, counter): mean value per dispatch for --pmc runs.
"""
import os
import sqlite3
import sys


def summarize(path):
    c = sqlite3.connect(path)
    print("# %s" % path)
    rows = c.execute("""
        select s.kernel_name, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start),
               max(d.end - d.start), max(s.arch_vgpr_count), max(s.accum_vgpr_count), max(s.sgpr_count),
               max(d.group_segment_size), max(d.private_segment_size), max(d.grid_size_x), max(d.workgroup_size_x)
        from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id
        group by s.kernel_name order by 3 desc""").fetchall()
    total = sum(r[2] for r in rows) or 1
    print("%-46s %6s %12s %12s %12s %12s %6s %5s %5s %5s %7s %7s %9s %5s" % (
        "kernel", "calls", "total_ms", "avg_ms", "min_ms", "max_ms", "pct", "vgpr", "agpr", "sgpr", "lds_B",
        "scr_B", "grid", "wg"))
    for r in rows:
        name = r[0].split("(")[0][-46:]
        print("%-46s %6d %12.4f %12.4f %12.4f %12.4f %6.2f %5s %5s %5s %7s %7s %9s %5s" % (
            name, r[1], r[2] / 1e6, r[3] / 1e6, r[4] / 1e6, r[5] / 1e6, 100.0 * r[2] / total, r[6], r[7], r[8],
            r[9], r[10], r[11], r[12]))
    try:
        pm = c.execute("""
            select s.kernel_name, p.name, count(*), avg(e.value), sum(e.value)
            from rocpd_pmc_event e join rocpd_info_pmc p on e.pmc_id = p.id
            join rocpd_kernel_dispatch d on d.event_id = e.event_id
            join rocpd_info_kernel_symbol s on d.kernel_id = s.id
            group by s.kernel_name, p.name order by s.kernel_name, p.name""").fetchall()
    except sqlite3.Error as ex:
        pm = []
        print("# (no pmc table: %s)" % ex)
    if pm:
        print("%-46s %-28s %8s %18s %18s" % ("kernel", "counter", "samples", "mean_per_dispatch", "sum"))
        for r in pm:
            print("%-46s %-28s %8d %18.1f %18.1f" % (r[0].split("(")[0][-46:], r[1], r[2], r[3], r[4]))
    # dispatch sequence of the scan kernels (sample pass / main pass alternate): last 12 dispatches
    seq = c.execute("""
        select s.kernel_name, d.end - d.start from rocpd_kernel_dispatch d
        join rocpd_info_kernel_symbol s on d.kernel_id = s.id
        where s.kernel_name like '%flat_scan%' order by d.start""").fetchall()
    if seq:
        print("# last scan-kernel dispatches in launch order (ms): " +
              " ".join("%.3f" % (r[1] / 1e6) for r in seq[-12:]))
    if os.environ.get("ROCPD_SEQ"):
        # every dispatch of the tail of the run in launch order: what one batch looks like
        n = int(os.environ["ROCPD_SEQ"])
        allseq = c.execute("""
            select s.kernel_name, d.start, d.end from rocpd_kernel_dispatch d
            join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start""").fetchall()
        prev_end = None
        print("# last %d dispatches in launch order: kernel, duration ms, gap to the previous kernel's end ms" % n)
        for name, st, en in allseq[-n:]:
            gap = (st - prev_end) / 1e6 if prev_end is not None else 0.0
            print("#   %-52s %9.4f  gap %8.4f" % (name.split("(")[0][-52:], (en - st) / 1e6, gap))
            prev_end = en
    print()


if __name__ == "__main__":
    import glob
    import os
    for p in sys.argv[1:]:
        if os.path.isdir(p):  # a rocprofv3 output directory: every *.db below it
            for f in sorted(glob.glob(os.path.join(p, "**", "*.db"), recursive=True)):
                summarize(f)
        else:
            summarize(p)
