#!/usr/bin/env python3
"""7x7 layers that leave through head_epilogue's raw fp32 NHWC path: the previous-frame stems on tile 61 (6 -> 32 at 2048x1024, 6 -> 64 at
1024x512, 6 -> 128 at 512x256) and the narrow label stem on tile 60 (108 -> 32 at 1024x512), conv + statistics, cold cache.
    python scripts/c8_bench.py > gpurun_out/c8_bench.txt"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn
from vid2vid_amd import lib as L
from vid2vid_amd.engine import Engine

DEV = "cuda:0"
eng = Engine(DEV, L.BF16)
THRASH = torch.empty(96 << 20, dtype=torch.float32, device=DEV)


def timed(fn, rounds=9):
    for _ in range(2):
        fn()
    ts = []
    for _ in range(rounds):
        THRASH.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[len(ts) // 2]


with torch.no_grad():
    for name, cin, cout, H, W, tiles in (("6->32 @2048x1024", 6, 32, 1024, 2048, (61, 13)), ("6->64 @1024x512", 6, 64, 512, 1024, (61, 13)),
                                         ("6->128 @512x256", 6, 128, 256, 512, (61, 14)), ("108->32 @1024x512", 108, 32, 512, 1024, (60, 120)),
                                         ("108->16 @1024x512", 108, 16, 512, 1024, (60,))):
        torch.manual_seed(0)
        conv = nn.Conv2d(cin, cout, 7).to(DEV)
        x = eng.pack(torch.randn(1, cin, H, W, device=DEV))
        if cin == 108:                                  # chunk-stride input as the frame hands it to the dense label stems
            wide = torch.zeros(1, H, W, 128, dtype=x.t.dtype, device=DEV)
            wide[..., :x.t.shape[-1]] = x.t
            from vid2vid_amd.engine import Act
            x = Act(wide, cin)
        out = []
        for t in tiles:
            eng.tile_override[(cin, cout, 7, 1, 0)] = (t, 1, 0)
            try:
                us = timed(lambda: eng.conv(x, conv, L.PAD_REFLECT, 3, L.OUT_RAW_F32_NHWC, want_stats=True))
                raw, rows, _ = eng.conv(x, conv, L.PAD_REFLECT, 3, L.OUT_RAW_F32_NHWC, want_stats=True)
                out.append("t%d: %.1f us (sum %.6e)" % (t, us, float(raw[:H * W * cout].double().sum())))
            except Exception as e:
                out.append("t%d: n/a (%s)" % (t, str(e)[:50]))
        print("%-20s | %s" % (name, "  ".join(out)), flush=True)
