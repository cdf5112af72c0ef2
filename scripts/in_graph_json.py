#!/usr/bin/env python3
"""Average IN-GRAPH duration of the dominant conv kernel from a rocprofv3 --kernel-trace database of the bench command
-> profiles/<tag>_in_graph.json (read by bench.py for roofline.in_graph).
    python scripts/in_graph_json.py <bench_results.db> <kernel-name substring> <tile,splitk,members> <out.json>"""
import json
import sqlite3
import sys

db, needle, cfg, out = sys.argv[1:5]
wgs = int(sys.argv[5]) if len(sys.argv) > 5 else 0      # only launches of this many workgroups (a template instance serves several layer shapes)
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)").fetchall()]
where = ""
if wgs:
    import re
    def col(kind, ax):
        pats = {"g": ("grid_size_%s", "grid_%s", "grid%s", "grid_size%s"), "w": ("workgroup_size_%s", "workgroup_%s", "wg_%s", "block_%s", "workgroup_size%s")}[kind]
        for p in pats:
            if p % ax in cols:
                return p % ax
        return None
    gx, gy, gz, wx, wy, wz = [col(k, a) for k in "gw" for a in "xyz"]
    if "grid_size" in cols and "workgroup_size" in cols:
        where = " and grid_size / workgroup_size = %d" % wgs
    elif all((gx, gy, gz, wx, wy, wz)):
        where = " and (%s / %s) * (%s / %s) * (%s / %s) = %d" % (gx, wx, gy, wy, gz, wz, wgs)
    else:
        print("in_graph_json: no grid / workgroup columns among", cols, file=sys.stderr)
rows = c.execute("select name, count(*), avg(end-start), min(end-start), max(end-start) from kernels where name like ?" + where +
                 " group by name order by 2 desc", ("%" + needle + "%",)).fetchall()
res = {"cfg": [int(v) for v in cfg.split(",")], "kernel": rows[0][0] if rows else None,
       "launches": rows[0][1] if rows else 0, "avg_us": round(rows[0][2] / 1e3, 2) if rows else None,
       "min_us": round(rows[0][3] / 1e3, 2) if rows else None, "max_us": round(rows[0][4] / 1e3, 2) if rows else None,
       "note": "rocprofv3 --kernel-trace of `python bench.py` replaying profiles/tune_cache.json: every launch of this kernel "
               "inside the frame hipGraph (3 lanes sharing the chip), warm-up and timed frames alike"
               + (("; launches of %d workgroups only (filter: %s)" % (wgs, where.strip() or "none: grid columns not found")) if wgs else "")}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res))
