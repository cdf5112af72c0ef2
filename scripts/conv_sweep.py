#!/usr/bin/env python3
"""Micro-benchmark of the implicit-GEMM conv template instances on the layer shapes that dominate the
512x256 frame.  HIP-event timing of back-to-back launches on one stream.

    python scripts/conv_sweep.py [bf16|fp32] > gpurun_out/conv_sweep.txt
"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn
from vid2vid_amd import lib as L
from vid2vid_amd.engine import Engine

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
eng = Engine("cuda:0", L.BF16 if prec == "bf16" else L.F32)
SHAPES = [
    # name, cin, cout, k, stride, pad, mode, H, W, transposed
    ("res1024 3x3 @32x64", 1024, 1024, 3, 1, 1, "reflect", 32, 64, False),
    ("res512 3x3 @32x64 (fg)", 512, 512, 3, 1, 1, "reflect", 32, 64, False),
    ("stem 108->128 7x7 @256x512", 108, 128, 7, 1, 3, "reflect", 256, 512, False),
    ("stem 108->64 7x7 @256x512", 108, 64, 7, 1, 3, "reflect", 256, 512, False),
    ("stem 6->128 7x7 @256x512", 6, 128, 7, 1, 3, "reflect", 256, 512, False),
    ("down 128->256 s2 @256x512", 128, 256, 3, 2, 1, "zero", 256, 512, False),
    ("down 256->512 s2 @128x256", 256, 512, 3, 2, 1, "zero", 128, 256, False),
    ("down 512->1024 s2 @64x128", 512, 1024, 3, 2, 1, "zero", 64, 128, False),
    ("up 1024->512 convT @32x64", 1024, 512, 3, 2, 1, "zero", 32, 64, True),
    ("up 512->256 convT @64x128", 512, 256, 3, 2, 1, "zero", 64, 128, True),
    ("up 256->128 convT @128x256", 256, 128, 3, 2, 1, "zero", 128, 256, True),
    ("head 128->3 7x7 @256x512", 128, 3, 7, 1, 3, "reflect", 256, 512, False),
    ("res128 3x3 @256x512 (scale1)", 128, 128, 3, 1, 1, "reflect", 256, 512, False),
    ("res64 3x3 @512x1024 (scale2)", 64, 64, 3, 1, 1, "reflect", 512, 1024, False),
]
REPS = 8
THRASH = torch.empty(96 << 20, dtype=torch.float32, device="cuda:0")
for name, cin, cout, k, stride, pad, mode, H, W, tr in SHAPES:
    if tr:
        mod = nn.ConvTranspose2d(cin, cout, k, stride=2, padding=pad, output_padding=1).to("cuda:0")
    else:
        mod = nn.Conv2d(cin, cout, k, stride=stride, padding=0 if mode == "reflect" else pad).to("cuda:0")
    x = eng.pack(torch.randn(1, cin, H, W, device="cuda:0"))
    pm = L.PAD_REFLECT if mode == "reflect" else L.PAD_ZERO
    res = []
    for tile in (0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17):
        key = (cin, cout, k, mod.stride[0], int(tr))
        eng.tile_override[key] = tile
        try:
            out_mode = L.OUT_RAW_F32_NHWC
            for _ in range(3):
                eng.conv(x, mod, pm, pad if not tr else None, out_mode, want_stats=True)
            torch.cuda.synchronize()
            tot = 0.0
            for _ in range(REPS):          # cold: a frame streams 0.7 GB of weights, nothing stays in L2 / MALL
                THRASH.zero_()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                eng.conv(x, mod, pm, pad if not tr else None, out_mode, want_stats=True)
                e1.record()
                e1.synchronize()
                tot += e0.elapsed_time(e1)
            us = tot * 1e3 / REPS
            fl = eng.conv_log[-1]["flops"]
            used = eng.conv_log[-1]["tile"]
            res.append((tile, used, us, fl / us / 1e6))
        except Exception as ex:      # a tile config may not fit a shape
            res.append((tile, -1, float("nan"), float("nan")))
    best = min((r for r in res if r[2] == r[2]), key=lambda r: r[2])
    print("%-34s %s | " % (name, prec) + "  ".join("t%d%s:%.1fus/%.0fTF" % (r[0], "(=%d)" % r[1] if r[0] == 0 else "", r[2], r[3]) for r in res)
          + "  | best t%d %.1f us" % (best[0], best[2]), flush=True)
