#!/usr/bin/env python3
"""Debug: tile search of a FlowNet2 deconvolution writing its own tensor vs a channel range of a wider concat buffer."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn as nn
from vid2vid_amd import lib as L
from vid2vid_amd.engine import Engine, Act
dev = torch.device("cuda:0")
eng = Engine(dev, L.BF16)
eng.autotune = True
torch.manual_seed(0)
m = nn.ConvTranspose2d(770, 128, 4, 2, 1).to(dev)
x = Act(torch.randn(3, 16, 32, 776, device=dev).bfloat16(), 770)
x.t[..., 770:] = 0
with torch.no_grad():
    for mode in ("own", "view"):
        eng._tuned.clear()
        out = None
        if mode == "view":
            buf = torch.zeros(3, 32, 64, 392, device=dev, dtype=torch.bfloat16)
            out = Act(buf[..., 256:384], 128)
        n0 = len(eng.conv_log)
        os.environ["V2V_TUNE_DEBUG"] = "1"
        res, _, _ = eng.conv(x, m, L.PAD_ZERO, None, L.OUT_ACT_NHWC, L.ACT_LEAKY, 0.1, 1.0, out=out, label="dbg")
        torch.cuda.synchronize()
        print(mode, "tuned:", dict(eng._tuned), "log tile:", eng.conv_log[-1]["tile"], eng.conv_log[-1]["splitk"])
        if mode == "own":
            ref = res.t.float().clone()
        else:
            print("max diff vs own:", (res.t.float() - ref).abs().max().item(), "outside range untouched:", float(buf[..., :256].abs().max()), float(buf[..., 384:].abs().max()))
