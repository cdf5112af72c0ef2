#!/usr/bin/env python3
"""GPU-side durations of scripts/fixed_cost.py's launches from a rocprofv3 --kernel-trace database (its own event timings bottom out
at the ~17 us host cost of one Python eng.conv call): conv dispatches in the script's order, 3 + 20 per (case, ablation).
    python scripts/fixed_cost_trace.py <results.db>"""
import sqlite3
import sys

CASES = ["down 128->256 @512x256 t18", "down 256->512 @256x128 t15", "down 512->1024 @128x64 t15/S2", "up 1024->512 @64x32 t13",
         "up 512->256 @128x64 t14", "up 256->128 @256x128 t14", "res 1024->1024 @64x32 t89", "res 1024->1024 @64x32 t89/S2",
         "2x res 1024->1024 paired t89"]
ABL = [0, 512, 1024, 1028]
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end from kernels where name like '%conv_igemm_kernel%' or name like '%conv3x3_pp3_kernel%' order by start").fetchall()
print("# kernel durations (us, rocprofv3 kernel trace, median of 20 warm launches); %d conv dispatches" % len(rows))
print("# ablate: 0 full kernel, 512 every workgroup returns at once, 1024 no main loop (prologue + epilogue), 1028 the same without output stores")
i = 0
for n in CASES:
    out = []
    for ab in ABL:
        d = sorted((e - s) / 1e3 for _, s, e in rows[i + 3:i + 23])
        i += 23
        if len(d) == 20:
            out.append("a%d:%.1f" % (ab, d[10]))
    print("%-32s %s" % (n, "  ".join(out)))
