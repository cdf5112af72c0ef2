#!/usr/bin/env python3
"""Per-kernel averages of the counters of one rocprofv3 `--pmc` pass (rocpd database) as a text table.
    python scripts/pmc_summary.py <results.db> [title...]"""
import sqlite3
import sys


def main(path, title=""):
    c = sqlite3.connect(path)
    rows = c.execute("select kernel_name, counter_name, count(*), sum(value), avg(value) from counters_collection "
                     "group by kernel_name, counter_name order by kernel_name, counter_name").fetchall()
    if title:
        print(title)
    print("# source: %s (rocprofv3 --pmc; per-dispatch averages)" % path)
    print("%-96s %-30s %8s %18s %18s" % ("kernel", "counter", "disp", "sum", "avg/dispatch"))
    for k, n, cnt, s, a in rows:
        print("%-96s %-30s %8d %18.1f %18.1f" % (k[:96], n, cnt, s, a))


if __name__ == "__main__":
    main(sys.argv[1], " ".join(sys.argv[2:]))
