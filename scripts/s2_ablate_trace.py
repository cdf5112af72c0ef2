#!/usr/bin/env python3
"""GPU-side durations of scripts/s2_ablate.py's launches from a rocprofv3 --kernel-trace database (the script's own event timings
are host-issue bound for these short kernels): the conv dispatches arrive in the script's order, 3 + 20 warm + 5 cold per
(shape, ablation).   python scripts/s2_ablate_trace.py <results.db>"""
import sqlite3
import sys

NAMES = ["down 128->256 @512x256 t18", "down 256->512 @256x128 t15", "down 512->1024 @128x64 t15/S2",
         "up 1024->512 @64x32 t13", "up 512->256 @128x64 t14", "up 256->128 @256x128 t14"]
ABL = [0, 1, 2, 3, 4, 7, 16, 19, 23]
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end from kernels where name like '%conv_igemm_kernel%' order by start").fetchall()
per = 28
print("# kernel durations (us, rocprofv3 kernel trace; median of 20 warm back-to-back / 5 cold launches), %d conv dispatches" % len(rows))
print("# ablate bits: 1 input rows from the zero page, 2 one hot weight chunk, 4 no stores, 16 loader only (no fragment reads, no MFMAs)")
i = 0
for n in NAMES:
    warm, cold = [], []
    for ab in ABL:
        d = [(e - s) / 1e3 for _, s, e in rows[i:i + per]]
        i += per
        if len(d) < per:
            break
        w = sorted(d[3:23]); k = sorted(d[23:28])
        warm.append("a%d:%.1f" % (ab, w[10])); cold.append("a%d:%.1f" % (ab, k[2]))
    print("%-30s warm  %s" % (n, "  ".join(warm)))
    print("%-30s cold  %s" % ("", "  ".join(cold)))
