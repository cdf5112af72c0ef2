#!/usr/bin/env python3
"""Kernel-stats summary of a rocprofv3 `--kernel-trace --stats` run.

rocprofv3 (ROCm 7.2) writes a rocpd SQLite database by default; this prints the same table
`--stats` would (per-kernel calls / total / average / min / max / share) so that the summary can
be committed under profiles/ as plain text.

    python scripts/rocprof_summary.py gpurun_out/prof/bench_results.db > profiles/rNN_<what>.txt
"""
import sqlite3
import sys


def main(path, header=""):
    c = sqlite3.connect(path)
    rows = c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), "
                     "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size) "
                     "from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    if header:
        print(header)
    print("# source: %s   (durations in microseconds; rocprofv3 kernel trace)" % path)
    print("%-100s %7s %12s %10s %10s %10s %6s %5s %5s %5s %7s" % (
        "kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct", "vgpr", "agpr", "sgpr", "lds"))
    for r in rows:
        print("%-100s %7d %12.1f %10.2f %10.2f %10.2f %6.2f %5d %5d %5d %7d" % (
            r[0][:100], r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / tot,
            r[6] or 0, r[7] or 0, r[8] or 0, r[9] or 0))


if __name__ == "__main__":
    main(sys.argv[1], " ".join(sys.argv[2:]))
