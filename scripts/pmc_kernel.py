#!/usr/bin/env python3
"""Average of one PMC counter for the kernels whose name contains <needle>, from a rocprofv3 --pmc database.
    python scripts/pmc_kernel.py <db> <COUNTER> <needle>"""
import sqlite3
import sys
db, counter, needle = sys.argv[1:4]
c = sqlite3.connect(db)
for name, n, v in c.execute("select kernel_name, count(*), avg(value) from counters_collection where counter_name = ? and kernel_name like ? "
                            "group by kernel_name", (counter, "%" + needle + "%")).fetchall():
    print("%s: %s avg %.1f over %d dispatches (%s)" % (counter, name[:90], v, n, "KiB; FETCH_SIZE is x2 on gfx950" if "SIZE" in counter else "raw"))
