#!/usr/bin/env python3
"""One steady-state batch of the int8 flat chain out of a rocprofv3 kernel trace (rocpd SQLite): every launch from
prep_queries_i8 to rerank256 with its duration, the gap to the previous kernel's end and its grid, plus the batch total.
usage: batch_timeline.py <dir with trace_results.db | file.db> [nth batch, default: the 6th from the end]"""
import glob
import os
import sqlite3
import sys


def main():
    p = sys.argv[1]
    if os.path.isdir(p):
        p = sorted(glob.glob(os.path.join(p, "**", "*.db"), recursive=True))[0]
    c = sqlite3.connect(p)
    seq = c.execute("""select s.kernel_name, d.start, d.end, d.grid_size_x, d.workgroup_size_x from rocpd_kernel_dispatch d
        join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start""").fetchall()
    batches = []
    i = 0
    while i < len(seq):
        if "prep_queries_i8" in seq[i][0]:
            j = i
            while j < len(seq) and j < i + 40 and "erank256" not in seq[j][0]:
                j += 1
            if j < len(seq) and "erank256" in seq[j][0]:
                batches.append(seq[i:j + 1])
                i = j
        i += 1
    if not batches:
        print("# no int8 batch in %s" % p)
        return
    nth = int(sys.argv[2]) if len(sys.argv) > 2 else max(0, len(batches) - 6)
    blk = batches[nth]
    print("# %s: batch %d of %d" % (p, nth, len(batches)))
    prev = None
    for b in blk:
        print("  %-64s dur %8.4f ms  gap %8.4f  workgroups %6d x %d" % (
            b[0].split("(")[0][-64:], (b[2] - b[1]) / 1e6, 0.0 if prev is None else (b[1] - prev) / 1e6, b[3] // max(b[4], 1), b[4]))
        prev = b[2]
    print("  total %.4f ms (first start to last end); sum of durations %.4f ms" % (
        (blk[-1][2] - blk[0][1]) / 1e6, sum(b[2] - b[1] for b in blk) / 1e6))
    tot = [(b[-1][2] - b[0][1]) / 1e6 for b in batches[len(batches) // 2:]]
    print("  batch totals of the second half of the run: mean %.4f min %.4f max %.4f ms (%d batches)" % (
        sum(tot) / len(tot), min(tot), max(tot), len(tot)))


if __name__ == "__main__":
    main()
