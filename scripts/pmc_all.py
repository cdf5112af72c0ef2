#!/usr/bin/env python3
"""Every PMC counter of a rocprofv3 --pmc database, averaged over the dispatches of the kernels whose name contains <needle>.
    python scripts/pmc_all.py <db> <needle> [label]"""
import sqlite3
import sys
db, needle = sys.argv[1:3]
label = sys.argv[3] if len(sys.argv) > 3 else ""
c = sqlite3.connect(db)
rows = c.execute("select counter_name, kernel_name, count(*), avg(value), min(value), max(value) from counters_collection "
                 "where kernel_name like ? group by counter_name, kernel_name order by kernel_name, counter_name", ("%" + needle + "%",)).fetchall()
if label:
    print("# " + label)
last = None
for cn, kn, n, av, mn, mx in rows:
    if kn != last:
        print("kernel %s (%d dispatches)" % (kn[:110], n))
        last = kn
    print("  %-34s avg %16.1f   min %16.1f   max %16.1f" % (cn, av, mn, mx))
