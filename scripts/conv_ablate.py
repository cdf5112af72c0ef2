#!/usr/bin/env python3
"""Where does the conv kernel's time go?  Ablations (v2v_conv_desc.ablate) and layout experiments on the dominant
layer shapes, cold cache (384 MB memset between launches).  Results of ablated launches are wrong by design.
    V2V_WPAD=0|1 python scripts/conv_ablate.py > gpurun_out/ablate.txt"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn
from vid2vid_amd import lib as L
from vid2vid_amd.engine import Engine, Act

eng = Engine("cuda:0", L.BF16)
SHAPES = [
    ("res1024 3x3 @32x64", 1024, 1024, 32, 64),
]
CFGS = [(54, 2, 0), (56, 2, 0), (51, 2, 0), (57, 2, 0), (52, 2, 0), (55, 2, 0)]
ABL = [0, 1, 2, 3, 4, 7, 16, 17, 18, 19, 23]
REPS = 7
THRASH = torch.empty(96 << 20, dtype=torch.float32, device="cuda:0")
print("V2V_WPAD=%s" % os.environ.get("V2V_WPAD", "(default)"))
for name, cin, cout, H, W in SHAPES:
    mod = nn.Conv2d(cin, cout, 3, padding=0).to("cuda:0")
    x0 = eng.pack(torch.randn(1, cin, H, W, device="cuda:0"))
    wide = torch.zeros(1, H, W, cin + 64, dtype=x0.t.dtype, device="cuda:0")
    wide[..., :cin] = x0.t
    xs = {"Cs=%d" % cin: x0}
    for xname, x in xs.items():
        for cfg in CFGS:
            eng.tile_override[(cin, cout, 3, 1, 0)] = cfg
            row = []
            for ab in ABL:
                eng.ablate = ab
                try:
                    for _ in range(2):
                        eng.conv(x, mod, L.PAD_REFLECT, 1, L.OUT_RAW_F32_NHWC, want_stats=True)
                    ts = []
                    for _ in range(REPS):
                        THRASH.zero_()
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        eng.conv(x, mod, L.PAD_REFLECT, 1, L.OUT_RAW_F32_NHWC, want_stats=True)
                        e1.record()
                        e1.synchronize()
                        ts.append(e0.elapsed_time(e1) * 1e3)
                    row.append("a%d:%.1f" % (ab, sorted(ts)[REPS // 2]))
                except Exception as ex:
                    row.append("a%d:ERR" % ab)
            eng.ablate = 0
            print("%-22s %-8s t%d/S%d/pf%d  %s" % (name, xname, cfg[0], cfg[1], cfg[2], "  ".join(row)), flush=True)
# launch-overhead reference: an empty-ish kernel timed the same way
e = torch.empty(1024, device="cuda:0")
ts = []
for _ in range(9):
    THRASH.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); e.zero_(); e1.record(); e1.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3)
print("event-bracketed tiny kernel: %.1f us" % sorted(ts)[4])
