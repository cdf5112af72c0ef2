#!/usr/bin/env python3
"""Round 6, VERDICT r5 item 6: the layer the V2V_STAMP_MASK build computes wrong (32 -> 32 3x3 at 8x16, fp32, tile 10 x split-K 2),
launched alone through the engine on whatever library V2V_LIB_PATH names.  Prints one line per (tile, split-K, statistics on/off):
max |diff| of the raw output against torch, relative to the output's rms; "nan" counts as wrong.

    V2V_LIB_PATH=build_stamp/lib_0x7f.so python scripts/stamp_probe.py          (on the GPU box)
"""
import os
import sys

import torch
import torch.nn as nn
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vid2vid_amd import lib as L                     # noqa: E402
from vid2vid_amd.engine import Engine                # noqa: E402

DEV = torch.device("cuda", 0)
torch.manual_seed(11)
eng = Engine(DEV, L.F32)
cin = cout = int(os.environ.get("PROBE_C", "32"))
H, W = 8, 16
conv = nn.Conv2d(cin, cout, 3, padding=0)
norm = nn.BatchNorm2d(cout).to(DEV)
xs = [torch.randn(1, cin, H, W) * (1.0 + i) for i in range(3)]
refs = [F.conv2d(F.pad(x, (1,) * 4, mode="reflect"), conv.weight.detach(), conv.bias.detach()) for x in xs]
conv = conv.to(DEV)
xa = [eng.pack(x.to(DEV)) for x in xs]
bad = 0
for it, (tile, S, stats) in enumerate([(t, s_, st) for t in (10, 9, 3) for s_ in (1, 2, 4) for st in (True, False)] * 2):
    k = it % 3
    eng.tile_override[(cin, cout, 3, 1, 0)] = (tile, S, 0)
    ss = torch.full((4 * cout,), float("nan"), device=DEV)
    try:
        if stats:
            raw, rows, (N, OH, OW) = eng.conv(xa[k], conv, L.PAD_REFLECT, 1, L.OUT_RAW_F32_NHWC, want_stats=True, fin=(norm, ss))
        else:
            raw, rows, (N, OH, OW) = eng.conv(xa[k], conv, L.PAD_REFLECT, 1, L.OUT_RAW_F32_NHWC, want_stats=False)
    except RuntimeError as e:
        print("tile %2d S %d stats %d: refused (%s)" % (tile, S, stats, str(e)[:60]))
        continue
    got = raw[:N * OH * OW * cout].view(N, OH, OW, cout).permute(0, 3, 1, 2).float().cpu()
    d = (got - refs[k]).abs()
    rel = float(d.max() / refs[k].pow(2).mean().sqrt())
    wrong = not (rel < 1e-4)
    bad += wrong
    where = ""
    if wrong:
        idx = (d > 1e-3 * refs[k].abs().max()).nonzero()
        where = " wrong elements %d of %d; channels %s rows %s cols %s" % (
            len(idx), d.numel(), sorted(set(idx[:, 1].tolist()))[:40], sorted(set(idx[:, 2].tolist())), sorted(set(idx[:, 3].tolist())))
    print("tile %2d S %d stats %d input %d: rel %.3e %s%s" % (tile, S, stats, k, rel, "WRONG" if wrong else "ok", where))
print("lib %s: %d wrong" % (os.path.basename(L.LIB_PATH), bad))
