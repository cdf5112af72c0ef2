#!/usr/bin/env python3
"""Launch ONE conv layer shape with a fixed (tile, split-K, prefetch) configuration `reps` times, cold cache between
launches -- the target of the rocprofv3 --pmc passes that measure the dominant kernel's HBM traffic per launch.
    python scripts/conv_layer_run.py --cfg 55,2,0 [--cin 1024 --cout 1024 --k 3 --H 32 --W 64] [--reps 20]"""
import argparse
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn
from vid2vid_amd import lib as L
from vid2vid_amd.engine import Engine

ap = argparse.ArgumentParser()
ap.add_argument("--cfg", default="55,2,0")
ap.add_argument("--cin", type=int, default=1024)
ap.add_argument("--cout", type=int, default=1024)
ap.add_argument("--k", type=int, default=3)
ap.add_argument("--H", type=int, default=32)
ap.add_argument("--W", type=int, default=64)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--batch", type=int, default=1, help="images per launch (2 = the geometry of a two-sequence launch: 256 workgroups of tile 90)")
ap.add_argument("--pair", action="store_true", help="paired launch (v2v_conv2d_pair) of two layers of this shape")
ap.add_argument("--fused", action="store_true", help="with --pair: norm + ReLU + residual inside the launch (V2V_OUT_NORM_ACT_NHWC), as the frame runs it")
a = ap.parse_args()
cfg = tuple(int(v) for v in a.cfg.split(","))
eng = Engine("cuda:0", L.BF16)
mod = nn.Conv2d(a.cin, a.cout, a.k, padding=0).to("cuda:0")
norm = nn.BatchNorm2d(a.cout).to("cuda:0")
x = eng.pack(torch.randn(a.batch, a.cin, a.H, a.W, device="cuda:0"))
ss = torch.zeros(4 * a.cout, device="cuda:0")
eng.tile_override[(a.cin, a.cout, a.k, 1, 0)] = cfg
thrash = torch.empty(96 << 20, dtype=torch.float32, device="cuda:0")
mod2 = nn.Conv2d(a.cin, a.cout, a.k, padding=0).to("cuda:0")
x2 = eng.pack(torch.randn(1, a.cin, a.H, a.W, device="cuda:0"))
ss2 = torch.zeros(4 * a.cout, device="cuda:0")
eng.pair_override = (cfg[0], cfg[1])
norm2 = nn.BatchNorm2d(a.cout).to("cuda:0")
r1 = eng.pack(torch.randn(1, a.cout, a.H, a.W, device="cuda:0"))
r2 = eng.pack(torch.randn(1, a.cout, a.H, a.W, device="cuda:0"))
with torch.no_grad():
    for _ in range(a.reps):
        thrash.zero_()
        if a.pair and a.fused:        # every second conv of a ResnetBlock pair: + residual (the first has ReLU and no residual)
            eng.conv_group_pair(x, mod, norm, x2, mod2, norm2, L.PAD_REFLECT, a.k // 2, L.ACT_NONE, 0.0,
                                adds_a=(r1, None), adds_b=(r2, None), labels=("a", "b"))
            assert eng.conv_log[-1].get("fused_norm")
        elif a.pair:
            eng.conv_pair(x, mod, x2, mod2, L.PAD_REFLECT, a.k // 2, ((norm, ss), (norm, ss2)), ("a", "b"))
        else:
            eng.conv(x, mod, L.PAD_REFLECT, a.k // 2, L.OUT_RAW_F32_NHWC, want_stats=True, fin=(norm, ss) if a.batch == 1 else None)
torch.cuda.synchronize()
print("ran %d launches of %s" % (a.reps, eng.conv_log[-1]))
