#!/usr/bin/env python3
"""The generator heads (7x7, 3 | 2+1 | 3 output channels, planar fp32 + activation) of the three scales: conv7x7_rowsum_kernel
(tile 62) beside conv7x7_head_kernel (tile 60), cold cache (384 MB memset between launches), bf16.
    python scripts/head_bench.py > gpurun_out/head_bench.txt"""
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


SHAPES = [("G0 128->3 @512x256", 128, 3, 256, 512), ("G0 fg 64->3 @512x256", 64, 3, 256, 512),
          ("G1 64->3 @1024x512", 64, 3, 512, 1024), ("G1 fg 32->3 @1024x512", 32, 3, 512, 1024),
          ("G2 32->3 @2048x1024", 32, 3, 1024, 2048), ("G2 fg 16->3 @2048x1024", 16, 3, 1024, 2048),
          ("C4 128->3 @512x512", 128, 3, 512, 512)]
with torch.no_grad():
    for name, cin, cout, H, W in SHAPES:
        mod = nn.Conv2d(cin, cout, 7).to("cuda:0")
        x = eng.pack(torch.randn(1, cin, H, W, device="cuda:0"))
        if x.Cs % 32 != 0:
            x = eng.widen(x, (x.Cs + 31) // 32 * 32)
        mb = (H * W * x.Cs * 2 + H * W * cout * 4) / 1e6
        out = []
        res = {}
        for cfg in [(60, 1, 0), (62, 1, 0)]:
            eng.tile_override[(cin, cout, 7, 1, 0)] = cfg
            us = timed(lambda: eng.conv(x, mod, L.PAD_REFLECT, 3, L.OUT_F32_NCHW, L.ACT_TANH, 0.0, 1.0))
            res[cfg[0]] = eng.conv(x, mod, L.PAD_REFLECT, 3, L.OUT_F32_NCHW, L.ACT_TANH, 0.0, 1.0)[0].clone()
            out.append("t%d: %7.1f us %6.0f GB/s" % (cfg[0], us, mb / us * 1e3 / 1e3))
        d = (res[60] - res[62]).abs().max().item()
        print("%-26s %7.1f MB | %s | max |t60 - t62| %.2e" % (name, mb, "   ".join(out), d), flush=True)
