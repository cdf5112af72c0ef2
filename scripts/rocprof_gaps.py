#!/usr/bin/env python3
"""Where the idle time of a training chunk sits (rocprofv3 --kernel-trace database): for every kernel of the steady part of the
trace, the gap between its start and the end of the previous kernel ON THE SAME QUEUE, summed per kernel that FOLLOWS the gap
and per (previous -> next) pair.  A serial chain of short dependent launches shows as a few microseconds behind every kernel;
a host stall (allocation, synchronisation, Python) as rare long gaps behind whatever happened to come next.

    python scripts/rocprof_gaps.py <results.db> > profiles/rNN_<what>_gaps.txt"""
import collections
import re
import sqlite3
import sys


def short(name):
    n = re.sub(r"^void ", "", name)
    n = re.sub(r"^_ZN3v2v\d+", "", n)
    n = n.replace("v2v::", "")
    n = re.split(r"[<(]|I[A-Z]", n)[0]
    return n[:44]


def main(path):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("PRAGMA table_info(kernels)").fetchall()]
    qcol = next((k for k in ("stream_id", "queue_id", "queue") if k in cols), None)
    rows = c.execute("select name, start, end, %s from kernels order by start" % (qcol or "0")).fetchall()
    t0 = rows[len(rows) // 2][1]                                 # second half of the trace: past start-up and tuning
    rows = [r for r in rows if r[1] >= t0]
    span = rows[-1][2] - rows[0][1]
    print("# source: %s ; queue column: %s ; kernels in the steady half: %d over %.1f ms" % (path, qcol, len(rows), span / 1e6))
    last = {}
    by_next, by_pair, hist = collections.Counter(), collections.Counter(), collections.Counter()
    cnt_next, cnt_pair = collections.Counter(), collections.Counter()
    per_q = collections.Counter()
    busy_q = collections.Counter()
    for name, s, e, q in rows:
        busy_q[q] += e - s
        per_q[q] += 1
        if q in last:
            pn, pe = last[q]
            gap = max(s - pe, 0)
            a, b = short(pn), short(name)
            by_next[b] += gap; cnt_next[b] += 1
            by_pair[(a, b)] += gap; cnt_pair[(a, b)] += 1
            hist["<2us" if gap < 2000 else "2-5us" if gap < 5000 else "5-10us" if gap < 10000 else "10-30us" if gap < 30000
                 else "30-100us" if gap < 100000 else ">=100us"] += gap
        last[q] = (name, e)
    for q in sorted(per_q, key=lambda k: -per_q[k]):
        print("# queue %s: %d kernels, busy %.1f ms" % (q, per_q[q], busy_q[q] / 1e6))
    tot = sum(by_next.values()) or 1
    print("# total same-queue gap: %.1f ms ; by gap length: %s" % (tot / 1e6, "  ".join("%s %.1f ms" % (k, v / 1e6) for k, v in hist.most_common())))
    print("%-46s %8s %10s %8s %6s" % ("gap in front of", "count", "total_us", "avg_us", "pct"))
    for k, v in by_next.most_common(30):
        print("%-46s %8d %10.1f %8.2f %6.2f" % (k, cnt_next[k], v / 1e3, v / 1e3 / cnt_next[k], 100.0 * v / tot))
    print()
    print("%-46s -> %-46s %8s %10s %8s" % ("previous", "next", "count", "total_us", "avg_us"))
    for (a, b), v in by_pair.most_common(40):
        print("%-46s -> %-46s %8d %10.1f %8.2f" % (a, b, cnt_pair[(a, b)], v / 1e3, v / 1e3 / cnt_pair[(a, b)]))


if __name__ == "__main__":
    main(sys.argv[1])
