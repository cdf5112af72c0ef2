#!/usr/bin/env python3
"""Dense 7x7 stems (the pooled label encodings: 108 input channels in a 128-channel stride): the 7x7-window tile 120 of the single-phase
patch kernel beside the generic / head tiles the cache selects today, cold cache, bf16, raw fp32 output + statistics rows.
    python scripts/stem7_bench.py > gpurun_out/stem7_bench.txt"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn
from vid2vid_amd import lib as L
from vid2vid_amd.engine import Engine

eng = Engine("cuda:0", L.BF16)
thrash = torch.empty(96 << 20, dtype=torch.float32, device="cuda:0")


def timed(run, reps=7):
    for _ in range(2):
        run()
    ts = []
    for _ in range(reps):
        thrash.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[len(ts) // 2]


SHAPES = [("G1.down_seg 108->64 @1024x512", 108, 64, 512, 1024, (10, 13, 14)), ("G1.indv_down 108->32 @1024x512", 108, 32, 512, 1024, (10, 60)),
          ("G0.down_seg 108->128 @512x256", 108, 128, 256, 512, (14, 10)), ("G0.indv_down 108->64 @512x256", 108, 64, 256, 512, (10, 14)),
          ("C4 stem 45->128 @512x512", 45, 128, 512, 512, (14, 10))]
with torch.no_grad():
    for name, cin, cout, H, W, others in SHAPES:
        mod = nn.Conv2d(cin, cout, 7).to("cuda:0")
        x = eng.pack(torch.randn(1, cin, H, W, device="cuda:0"))
        if x.Cs % 64 != 0:
            x = eng.widen(x, (x.Cs + 63) // 64 * 64)
        out, ref = [], None
        for t in tuple(others) + (120, 121):
            eng.tile_override[(cin, cout, 7, 1, 0)] = (t, 1, 0)
            try:
                us = timed(lambda: eng.conv(x, mod, L.PAD_REFLECT, 3, L.OUT_RAW_F32_NHWC, want_stats=True))
            except Exception as e:
                out.append("t%d: n/a" % t); continue
            raw = eng.conv(x, mod, L.PAD_REFLECT, 3, L.OUT_RAW_F32_NHWC, want_stats=True)[0][:H * W * cout].clone()
            ref = raw if ref is None else ref
            gf = 2.0 * H * W * cin * cout * 49 / 1e9
            out.append("t%d: %6.1f us %5.0f TF (d %.1e)" % (t, us, gf / us * 1e3, (raw - ref).abs().max().item()))
        print("%-32s Cs %3d | %s" % (name, x.Cs, "  ".join(out)), flush=True)
