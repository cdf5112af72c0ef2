#!/usr/bin/env python3
"""One-hot (gather-sum) 7x7 stems as the frames run them: 108 -> 128 / 64 at 512x256 and 108 -> 32 / 16 at 2048x1024, uint8 label | edge
codes, cold cache (384 MB memset between launches), conv + statistics + in-kernel finalize.  For A/B of two builds on one box:
    V2V_LIB_PATH=<other build> python scripts/onehot_ab.py"""
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
T, nc = 3, 35
torch.manual_seed(0)


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


out = []
with torch.no_grad():
    for H, W, couts in ((256, 512, (128, 64)), (1024, 2048, (32, 16))):
        lab = torch.randint(0, nc, (T, H // 16 + 1, W // 16 + 1), device=DEV).repeat_interleave(16, 1).repeat_interleave(16, 2)[:, :H, :W].to(torch.uint8).contiguous()
        inst = torch.randint(0, 20, (T, H // 16 + 1, W // 16 + 1), device=DEV).repeat_interleave(16, 1).repeat_interleave(16, 2)[:, :H, :W].to(torch.int32).contiguous()
        x, _ = eng.encode_labels(lab, inst, T, H, W, nc, (), False)
        eng.label_codes(x.onehot, H, W)
        for cout in couts:
            conv = nn.Conv2d(T * (nc + 1), cout, 7).to(DEV)
            norm = nn.BatchNorm2d(cout).to(DEV)
            ss = torch.zeros(4 * cout, device=DEV)
            us = timed(lambda: eng.onehot_conv(x, conv, label="stem", fin=(norm, ss)))
            out.append("108->%d @%dx%d: %.1f us" % (cout, W, H, us))
print(os.environ.get("V2V_LIB_PATH", "tree build"), "|", "  ".join(out), flush=True)
