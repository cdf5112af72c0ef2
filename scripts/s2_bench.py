#!/usr/bin/env python3
"""The stride-2 3x3 layers of the 512x256 frame (+ the 2048x1024 fine scales): conv3x3_s2_kernel tiles 100-103 beside the generic
implicit-GEMM tiles the tile search selects today, cold cache (384 MB memset between launches), conv + statistics + in-kernel finalize.
    python scripts/s2_bench.py > gpurun_out/s2_bench.txt"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn
from vid2vid_amd import lib as L
from vid2vid_amd.engine import Engine

eng = Engine("cuda:0", L.BF16)
thrash = torch.empty(96 << 20, dtype=torch.float32, device="cuda:0")


def timed(run, reps=9):
    for _ in range(2):
        run()
    ts = []
    for _ in range(reps):
        thrash.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[len(ts) // 2]


TSHAPES = [("up 1024->512 @64x32", 1024, 512, 32, 64), ("up 512->256 @128x64", 512, 256, 64, 128), ("up 256->128 @256x128", 256, 128, 128, 256),
           ("fg up 512->256 @64x32", 512, 256, 32, 64), ("fg up 256->128 @128x64", 256, 128, 64, 128), ("fg up 128->64 @256x128", 128, 64, 128, 256),
           ("s1 up 128->64 @512x256", 128, 64, 256, 512), ("s2 up 64->32 @1024x512", 64, 32, 512, 1024), ("s1 fg up 64->32 @512x256", 64, 32, 256, 512),
           ("s2 fg up 32->16 @1024x512", 32, 16, 512, 1024)]      # 64-byte pixels: tile 114 through the paired-x view (engine.PairedXConvT)
SHAPES = [("down 128->256 @512x256", 128, 256, 256, 512), ("down 256->512 @256x128", 256, 512, 128, 256),
          ("down 512->1024 @128x64", 512, 1024, 64, 128), ("fg 64->128 @512x256", 64, 128, 256, 512),
          ("fg 128->256 @256x128", 128, 256, 128, 256), ("fg 256->512 @128x64", 256, 512, 64, 128),
          ("s1 64->128 @1024x512", 64, 128, 512, 1024)]
GENERIC = [(14, 1, 0), (15, 1, 0), (17, 1, 0), (18, 1, 0), (13, 1, 0), (15, 2, 0), (10, 1, 0)]
with torch.no_grad():
    for name, cin, cout, H, W in ([] if os.environ.get("T2_ONLY") else SHAPES):
        mod = nn.Conv2d(cin, cout, 3, stride=2, padding=1).to("cuda:0")
        norm = nn.BatchNorm2d(cout).to("cuda:0")
        ss = torch.zeros(4 * cout, device="cuda:0")
        x = eng.pack(torch.randn(1, cin, H, W, device="cuda:0"))
        gf = 2.0 * (H // 2) * (W // 2) * cout * cin * 9 / 1e9
        out = []
        for cfg in GENERIC + [(100, 1, 0), (101, 1, 0), (102, 1, 0), (103, 1, 0)]:
            eng.tile_override[(cin, cout, 3, 2, 0)] = cfg
            try:
                fin = (norm, ss) if (H // 2) * (W // 2) <= 32768 else None
                us = timed(lambda: eng.conv(x, mod, L.PAD_ZERO, None, L.OUT_RAW_F32_NHWC, want_stats=True, fin=fin))
                out.append("t%d%s:%.1f" % (cfg[0], "/S%d" % cfg[1] if cfg[1] > 1 else "", us))
            except Exception as ex:
                out.append("t%d:-" % cfg[0])
        print("%-26s %6.2f GF | %s" % (name, gf, "  ".join(out)), flush=True)
    for name, cin, cout, H, W in TSHAPES:
        mod = nn.ConvTranspose2d(cin, cout, 3, stride=2, padding=1, output_padding=1).to("cuda:0")
        norm = nn.BatchNorm2d(cout).to("cuda:0")
        ss = torch.zeros(4 * cout, device="cuda:0")
        x = eng.pack(torch.randn(1, cin, H, W, device="cuda:0"))
        gf = 2.0 * H * W * cout * cin * 9 / 1e9
        out = []
        for cfg in [(4, 1, 0), (13, 1, 0), (14, 1, 0), (17, 1, 0), (10, 1, 0), (9, 1, 0), (110, 1, 0), (111, 1, 0), (112, 1, 0), (113, 1, 0), (114, 1, 0)]:
            if os.environ.get("T2_ONLY") and cfg[0] < 110 and not (cfg[0] == 4 and cout <= 32):
                continue
            eng.tile_override[(cin, cout, 3, 2, 1)] = cfg
            try:
                fin = (norm, ss) if (4 * H * W <= 32768 or cfg[0] == 114) else None
                us = timed(lambda: eng.conv(x, mod, L.PAD_ZERO, None, L.OUT_RAW_F32_NHWC, want_stats=True, fin=fin))
                out.append("t%d:%.1f" % (cfg[0], us))
            except Exception as ex:
                out.append("t%d:-" % cfg[0])
        print("%-26s %6.2f GF | %s" % (name, gf, "  ".join(out)), flush=True)
