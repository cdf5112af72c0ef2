#!/usr/bin/env python3
"""Where does the time of the stride-2 / transposed layers go on the generic implicit-GEMM tiles?  Ablations (v2v_conv_desc.ablate:
1 input rows from the zero page, 2 one hot weight chunk, 4 no stores, 16 loader only -- no fragment reads / MFMAs) on the six
512x256-frame shapes with the tiles the committed cache selects; WARM back-to-back launches (REPS between two events, as inside the
frame graph) and single cold launches.  Results of ablated launches are wrong by design.
    python scripts/s2_ablate.py > gpurun_out/s2_ablate.txt"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn
from vid2vid_amd import lib as L
from vid2vid_amd.engine import Engine

eng = Engine("cuda:0", L.BF16)
#        name                         cin   cout  H    W   transposed  (tile, splitk, prefetch)
SHAPES = [("down 128->256 @512x256",   128,  256, 256, 512, 0, (18, 1, 0)),
          ("down 256->512 @256x128",   256,  512, 128, 256, 0, (15, 1, 0)),
          ("down 512->1024 @128x64",   512, 1024,  64, 128, 0, (15, 2, 0)),
          ("up 1024->512 @64x32",     1024,  512,  32,  64, 1, (13, 1, 0)),
          ("up 512->256 @128x64",      512,  256,  64, 128, 1, (14, 1, 0)),
          ("up 256->128 @256x128",     256,  128, 128, 256, 1, (14, 1, 0))]
ABL = [0, 1, 2, 3, 4, 7, 16, 19, 23]
REPS = 20
FIN = os.environ.get("FIN", "1") == "1"        # in-kernel norm finalize by the last workgroup of a channel tile
STATS = os.environ.get("STATS", "1") == "1"    # per-tile (sum, sum^2) rows
print("FIN=%d STATS=%d" % (FIN, STATS))
thrash = torch.empty(96 << 20, dtype=torch.float32, device="cuda:0")
with torch.no_grad():
    for name, cin, cout, H, W, tr, cfg in SHAPES:
        mod = (nn.ConvTranspose2d(cin, cout, 3, stride=2, padding=1, output_padding=1) if tr
               else nn.Conv2d(cin, cout, 3, stride=2, padding=1)).to("cuda:0")
        norm = nn.BatchNorm2d(cout).to("cuda:0")
        ss = torch.zeros(4 * cout, device="cuda:0")
        x = eng.pack(torch.randn(1, cin, H, W, device="cuda:0"))
        eng.tile_override[(cin, cout, 3, 2, tr)] = cfg
        warm, cold = [], []
        for ab in ABL:
            eng.ablate = ab
            run = lambda: eng.conv(x, mod, L.PAD_ZERO, None, L.OUT_RAW_F32_NHWC, want_stats=STATS, fin=(norm, ss) if FIN else None)
            for _ in range(3):
                run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(REPS):
                run()
            e1.record(); e1.synchronize()
            warm.append("a%d:%.1f" % (ab, e0.elapsed_time(e1) * 1e3 / REPS))
            ts = []
            for _ in range(5):
                thrash.zero_()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); run(); e1.record(); e1.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            cold.append("a%d:%.1f" % (ab, sorted(ts)[2]))
        eng.ablate = 0
        c = eng.conv_log[-1]
        print("%-24s tile %s  grid %s" % (name, cfg, {k: c.get(k) for k in ("tile", "splitk")}))
        print("   warm x%d us/launch  %s" % (REPS, "  ".join(warm)))
        print("   cold single us      %s" % "  ".join(cold), flush=True)
