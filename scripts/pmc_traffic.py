#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE of the dominant conv kernel (two separate rocprofv3 --pmc passes over
scripts/conv_layer_run.py) -> profiles/<tag>_traffic.json, read by bench.py for roofline.traffic.
FETCH_SIZE is doubled (gfx950 rocprofv3 reports half of the bytes of wide coalesced reads, MI355X_MICROARCH.md HBM
section); WRITE_SIZE is taken as reported (uncalibrated).  Units of both counters: KiB.
    python scripts/pmc_traffic.py <fetch.db> <write.db> <tile,splitk,members> <out.json>"""
import json
import sqlite3
import sys


def avg(path, counter):
    c = sqlite3.connect(path)
    rows = c.execute("select kernel_name, count(*), avg(value) from counters_collection where counter_name = ? "
                     "group by kernel_name", (counter,)).fetchall()
    rows = [r for r in rows if "conv3x3" in r[0] or "conv_igemm" in r[0] or "conv7x7" in r[0]]
    rows.sort(key=lambda r: -r[1])
    return rows[0] if rows else None


f, w, cfg, out = sys.argv[1:5]
fr, wr = avg(f, "FETCH_SIZE"), avg(w, "WRITE_SIZE")
res = {"cfg": [int(v) for v in cfg.split(",")], "kernel": fr[0] if fr else None,
       "fetch_kib_per_launch_reported": fr[2] if fr else None, "write_kib_per_launch_reported": wr[2] if wr else None,
       "dispatches": fr[1] if fr else 0,
       "hbm_bytes_per_launch": (2.0 * fr[2] + wr[2]) * 1024.0 if fr and wr else None,
       "note": "rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in separate passes over scripts/conv_layer_run.py (cold cache, "
               "res1024 3x3 layer: 1024->1024 channels at 32x64 pixels); FETCH_SIZE x2 (gfx950 correction), WRITE_SIZE as reported"}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res))
