#!/usr/bin/env python3
"""Per-(kernel, grid) table of a rocprofv3 --kernel-trace database: the same kernel on different layer shapes shows as
different grids, so the table says WHICH layers of the training step a kernel's time belongs to.  Also prints how busy
the device was over the trace's steady part (union of kernel intervals / span).

    python scripts/rocprof_by_grid.py <results.db> [name-substring ...] > profiles/rNN_<what>_by_grid.txt"""
import sqlite3
import sys


def main(path, pats):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("PRAGMA table_info(kernels)").fetchall()]
    gcols = [k for k in ("grid_x", "grid_y", "grid_z", "grid_size_x", "grid_size_y", "grid_size_z") if k in cols]
    wcols = [k for k in ("workgroup_x", "workgroup_size_x") if k in cols]
    sel = ", ".join(gcols + wcols) or "0"
    rows = c.execute("select name, %s, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels "
                     "group by name, %s order by sum(end-start) desc" % (sel, sel)).fetchall()
    tot = sum(r[-4] for r in rows) or 1
    iv = c.execute("select start, end from kernels order by start").fetchall()
    t0, t1 = iv[len(iv) // 2][0], iv[-1][1]                      # second half of the trace: past start-up and tuning
    busy, cur_s, cur_e = 0, None, None
    for s, e in iv:
        if e <= t0:
            continue
        s = max(s, t0)
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += (cur_e - cur_s) if cur_e is not None else 0
    print("# source: %s ; columns of the grid: %s ; workgroup: %s" % (path, gcols, wcols))
    print("# device busy over the second half of the trace: %.1f ms of %.1f ms = %.3f (union of kernel intervals / span)"
          % (busy / 1e6, (t1 - t0) / 1e6, busy / max(t1 - t0, 1)))
    # per training chunk: the largest Adam launch (G's flat buffer) ends a chunk's G phase -- the intervals between two of them
    # are whole chunks; busy = union of kernel intervals inside
    if gcols:
        ad = c.execute("select start, end, %s from kernels where name like '%%adam_step%%' order by start" % gcols[0]).fetchall()
        if ad:
            gmax = max(r[2] for r in ad)
            marks = [r[1] for r in ad if r[2] == gmax]
            rows_c = []
            for a, b in zip(marks[:-1], marks[1:]):
                busy_c, ce, n_k = 0, None, 0
                for s_, e_ in iv:
                    if e_ <= a or s_ >= b:
                        continue
                    s2, e2 = max(s_, a), min(e_, b)
                    n_k += 1
                    if ce is None or s2 > ce:
                        busy_c += e2 - s2
                        ce = e2
                    elif e2 > ce:
                        busy_c += e2 - ce
                        ce = e2
                rows_c.append(((b - a) / 1e6, busy_c / 1e6, n_k))
            print("# chunks (between consecutive G-optimizer Adam launches): span ms / busy ms / kernels: "
                  + "  ".join("%.1f/%.1f/%d" % r for r in rows_c[-12:]))
    print("%-72s %-22s %7s %11s %9s %9s %9s %6s" % ("kernel", "grid/wg", "calls", "total_us", "avg_us", "min_us", "max_us", "pct"))
    n = 0
    for r in rows:
        name = r[0]
        if pats and not any(p in name for p in pats):
            continue
        g = "x".join(str(v) for v in r[1:1 + len(gcols) + len(wcols)])
        print("%-72s %-22s %7d %11.1f %9.2f %9.2f %9.2f %6.2f" % (name[:72], g, r[-5], r[-4] / 1e3, r[-3] / 1e3, r[-2] / 1e3, r[-1] / 1e3, 100.0 * r[-4] / tot))
        n += 1
        if n >= 90:
            break


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2:])
