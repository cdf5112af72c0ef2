#!/usr/bin/env python3
"""Sweep (tile config, split-K, weight prefetch) on the layer shapes that dominate the 512x256 frame, cold
(384 MB memset between launches: weights come from HBM, as in a real frame) and warm (back-to-back).
    python scripts/conv_sweep2.py [bf16|fp32] > gpurun_out/conv_sweep2.txt"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
import torch.nn as nn
from vid2vid_amd import lib as L
from vid2vid_amd.lib import lib
from vid2vid_amd.engine import Engine, TILE_CFGS, PATCH_CFGS, PREFETCH_DIST, _stream

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
eng = Engine("cuda:0", L.BF16 if prec == "bf16" else L.F32)
SHAPES = [
    ("res1024 3x3 @32x64", 1024, 1024, 3, 1, 1, "reflect", 32, 64, False),
    ("res512 3x3 @32x64 (fg)", 512, 512, 3, 1, 1, "reflect", 32, 64, False),
    ("res128 3x3 @256x512 (scale1)", 128, 128, 3, 1, 1, "reflect", 256, 512, False),
]
REPS = 5
THRASH = torch.empty(96 << 20, dtype=torch.float32, device="cuda:0")
for name, cin, cout, k, stride, pad, mode, H, W, tr in SHAPES:
    if tr:
        mod = nn.ConvTranspose2d(cin, cout, k, stride=2, padding=pad, output_padding=1).to("cuda:0")
    else:
        mod = nn.Conv2d(cin, cout, k, stride=stride, padding=0 if mode == "reflect" else pad).to("cuda:0")
    x = eng.pack(torch.randn(1, cin, H, W, device="cuda:0"))
    pm = L.PAD_REFLECT if mode == "reflect" else L.PAD_ZERO
    key = (cin, cout, k, mod.stride[0], int(tr))
    M = (H * W) if tr else ((H + 2 * pad - k) // stride + 1) * ((W + 2 * pad - k) // stride + 1)
    ncls = 4 if tr else 1
    cands = []
    for t, (bm, bn, helper) in sorted(TILE_CFGS.items()):
        if t not in (13, 14, 15, 17):
            continue
        tiles = -(-M // (bm * ncls)) * -(-cout // bn) * ncls
        for S in (1, 2, 3, 4, 6, 8):
            if S > 1 and tiles * S > 1024:
                continue
            cands.append((t, S, 0))
            if helper:
                cands.append((t, S, PREFETCH_DIST))
                if S == 1:
                    cands.append((t, S, 24))
    if not tr and k == 3 and stride == 1:
        for t_ in sorted(PATCH_CFGS):
            for S in (1, 2, 3, 4, 8):
                cands.append((t_, S, 0))
                if t_ <= 37:
                    cands.append((t_, S, 12)); cands.append((t_, S, 24))
    res = []
    for cfg in cands:
        eng.tile_override[key] = cfg
        try:
            for _ in range(2):
                eng.conv(x, mod, pm, pad if not tr else None, L.OUT_RAW_F32_NHWC, want_stats=True)
            torch.cuda.synchronize()
            out = {}
            for cold in (True, False):
                ts = []
                for _ in range(REPS):
                    if cold:
                        THRASH.zero_()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    eng.conv(x, mod, pm, pad if not tr else None, L.OUT_RAW_F32_NHWC, want_stats=True)
                    e1.record()
                    e1.synchronize()
                    ts.append(e0.elapsed_time(e1) * 1e3)
                out[cold] = sorted(ts)[REPS // 2]
            fl = eng.conv_log[-1]["flops"]
            res.append((cfg, out[True], out[False], fl))
        except Exception as ex:
            pass
    res.sort(key=lambda r: r[1])
    fl = res[0][3]
    print("== %s %s  (%.2f GFLOP)" % (name, prec, fl / 1e9))
    print("   best cold: " + "  ".join("t%d/S%d/pf%d:%.1f(%.1fw)us=%.0fTF" % (c[0], c[1], c[2], a, b, f / a / 1e6) for c, a, b, f in res[:12]))
    base = [r for r in res if r[0][1] == 1 and r[0][2] == 0]
    print("   no split/pf: " + "  ".join("t%d:%.1f(%.1fw)" % (c[0], a, b) for c, a, b, f in sorted(base, key=lambda r: r[0][0])))
    pf = [r for r in res if r[0][1] == 1 and r[0][2] > 0]
    print("   prefetch only: " + "  ".join("t%d/pf%d:%.1f(%.1fw)" % (c[0], c[2], a, b) for c, a, b, f in sorted(pf, key=lambda r: r[0])))
    sys.stdout.flush()
