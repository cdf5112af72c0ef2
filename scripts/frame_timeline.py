#!/usr/bin/env python3
"""Kernel timeline of ONE frame of the bench command from a rocprofv3 --kernel-trace database: every kernel of the last
complete hipGraph replay with its start / end relative to the frame's first kernel (encode_labels), so the critical path and
the idle gaps of the frame graph can be read off.
    python scripts/frame_timeline.py <bench_results.db> > gpurun_out/<tag>_frame_timeline.txt"""
import re
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end from kernels order by start").fetchall()
starts = [i for i, r in enumerate(rows) if "encode_labels" in r[0]]
if len(starts) < 3:
    print("no frames found"); sys.exit(0)
a, b = starts[-2], starts[-1]
t0 = rows[a][1]
print("# one frame = kernels between the last two encode_labels launches; times in us relative to the first kernel")
print("# frame span: %.1f us, %d kernels, sum of kernel durations %.1f us" % (
    (rows[b][1] - t0) / 1e3, b - a, sum(r[2] - r[1] for r in rows[a:b]) / 1e3))


def short(n):
    m = re.search(r"v2v\d*(\w+?)(I|E|$)", n)
    n = re.sub(r"^_ZN3v2v\d+", "", n)
    n = re.sub(r"EvNS_.*$", "", n)
    n = n.replace("DF16b", "bf16,").replace("Li", "").replace("ELb0E", "").replace("E", ",")
    return n[:70]


busy_until = 0
for name, s, e in rows[a:b]:
    gap = (s - busy_until) / 1e3 if busy_until else 0.0
    print("%9.1f %9.1f %8.1f  %s%s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, short(name),
                                       "   <-- chip idle %.1f us before" % gap if gap > 3.0 else ""))
    busy_until = max(busy_until, e)
