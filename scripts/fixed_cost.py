#!/usr/bin/env python3
"""Fixed cost per conv launch: the launch itself (ablate 512: every workgroup returns at once), prologue + epilogue without a main
loop (1024), the same without stores (1028), against the full kernel (0).  Warm back-to-back launches (REPS between two events).
Generic implicit-GEMM tiles on the stride-2 / transposed shapes of the 512x256 frame, and the single-phase 3x3 kernel (ablation
instance 89 of tile 80) alone, split-K 2, and as a paired launch (the ablation instance has the raw-output epilogue, not the fused norm).
    python scripts/fixed_cost.py > gpurun_out/fixed_cost.txt"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn
from vid2vid_amd import lib as L
from vid2vid_amd.engine import Engine

eng = Engine("cuda:0", L.BF16)
ABL = [0, 512, 1024, 1028]
FIN = os.environ.get("FIN", "1") == "1"        # in-kernel norm finalize
STATS = os.environ.get("STATS", "1") == "1"    # per-tile statistics rows
REPS = 20


def timed(run):
    out = []
    for ab in ABL:
        eng.ablate = ab
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(REPS):
            run()
        e1.record(); e1.synchronize()
        out.append("a%d:%.1f" % (ab, e0.elapsed_time(e1) * 1e3 / REPS))
    eng.ablate = 0
    return "  ".join(out)


SHAPES = [("down 128->256 @512x256",   128,  256, 256, 512, 0, (18, 1, 0)),
          ("down 256->512 @256x128",   256,  512, 128, 256, 0, (15, 1, 0)),
          ("down 512->1024 @128x64",   512, 1024,  64, 128, 0, (15, 2, 0)),
          ("up 1024->512 @64x32",     1024,  512,  32,  64, 1, (13, 1, 0)),
          ("up 512->256 @128x64",      512,  256,  64, 128, 1, (14, 1, 0)),
          ("up 256->128 @256x128",     256,  128, 128, 256, 1, (14, 1, 0))]
with torch.no_grad():
    for name, cin, cout, H, W, tr, cfg in SHAPES:
        mod = (nn.ConvTranspose2d(cin, cout, 3, stride=2, padding=1, output_padding=1) if tr
               else nn.Conv2d(cin, cout, 3, stride=2, padding=1)).to("cuda:0")
        norm = nn.BatchNorm2d(cout).to("cuda:0")
        ss = torch.zeros(4 * cout, device="cuda:0")
        x = eng.pack(torch.randn(1, cin, H, W, device="cuda:0"))
        eng.tile_override[(cin, cout, 3, 2, tr)] = cfg
        print("%-26s tile %-12s %s" % (name, cfg, timed(lambda: eng.conv(x, mod, L.PAD_ZERO, None, L.OUT_RAW_F32_NHWC, want_stats=STATS, fin=(norm, ss) if FIN else None))), flush=True)
    cin = cout = 1024
    H, W = 32, 64
    mods = [nn.Conv2d(cin, cout, 3, padding=0).to("cuda:0") for _ in range(2)]
    norms = [nn.BatchNorm2d(cout).to("cuda:0") for _ in range(2)]
    xs = [eng.pack(torch.randn(1, cin, H, W, device="cuda:0")) for _ in range(2)]
    rs = [eng.pack(torch.randn(1, cout, H, W, device="cuda:0")) for _ in range(2)]
    ss = torch.zeros(4 * cout, device="cuda:0")
    for S in (1, 2):
        eng.tile_override[(cin, cout, 3, 1, 0)] = (89, S, 0)
        print("%-26s tile %-12s %s" % ("res 1024->1024 @64x32", (89, S, 0),
              timed(lambda: eng.conv(xs[0], mods[0], L.PAD_REFLECT, 1, L.OUT_RAW_F32_NHWC, want_stats=STATS, fin=(norms[0], ss) if FIN else None))), flush=True)
    eng.pair_override = (89, 1)
    ss2 = torch.zeros(4 * cout, device="cuda:0")
    print("%-26s tile %-12s %s" % ("2x res 1024->1024 paired", (89, 1, "pair"),
          timed(lambda: eng.conv_pair(xs[0], mods[0], xs[1], mods[1], L.PAD_REFLECT, 1, ((norms[0], ss), (norms[1], ss2)), ("a", "b")))), flush=True)
    print("last pair launch:", {k: eng.conv_log[-1].get(k) for k in ("tile", "splitk", "fused_norm")})
