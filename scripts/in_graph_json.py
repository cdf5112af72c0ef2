#!/usr/bin/env python3
"""Average IN-GRAPH duration of the dominant conv kernel from a rocprofv3 --kernel-trace database of the bench command
-> profiles/<tag>_in_graph.json (read by bench.py for roofline.in_graph).
    python scripts/in_graph_json.py <bench_results.db> <kernel-name substring> <tile,splitk,members> <out.json>"""
import json
import sqlite3
import sys

db, needle, cfg, out = sys.argv[1:5]
c = sqlite3.connect(db)
rows = c.execute("select name, count(*), avg(end-start), min(end-start), max(end-start) from kernels where name like ? "
                 "group by name order by 2 desc", ("%" + needle + "%",)).fetchall()
res = {"cfg": [int(v) for v in cfg.split(",")], "kernel": rows[0][0] if rows else None,
       "launches": rows[0][1] if rows else 0, "avg_us": round(rows[0][2] / 1e3, 2) if rows else None,
       "min_us": round(rows[0][3] / 1e3, 2) if rows else None, "max_us": round(rows[0][4] / 1e3, 2) if rows else None,
       "note": "rocprofv3 --kernel-trace of `python bench.py` replaying profiles/tune_cache.json: every launch of this kernel "
               "inside the frame hipGraph (3 lanes sharing the chip), warm-up and timed frames alike"}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res))
