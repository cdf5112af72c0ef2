#!/usr/bin/env python3
"""Where do the 100 us of a fine-scale ResnetBlock convolution go?  64 -> 64 3x3 at 1024x512 (2048 tiles of 256 px x 64, ONE channel
chunk = 9 tap steps per tile) on the instrumented single-phase tile 89 (csrc/conv3x3_pp3_kernel.h, ABL = 1): each ablation removes
one ingredient (results are wrong, only the time is meaningful).  Cold cache (384 MB memset between launches), median of 9.
    python scripts/fine_ablate.py > gpurun_out/fine_ablate.txt"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn
from vid2vid_amd import lib as L
from vid2vid_amd.engine import Engine

DEV = "cuda:0"
eng = Engine(DEV, L.BF16)
thrash = torch.empty(96 << 20, dtype=torch.float32, device=DEV)
ABL = [(0, "full kernel"), (512, "every workgroup returns at once (launch + dispatch of the grid)"), (1024, "no main loop (prologue + epilogue)"),
       (1024 | 4, "no main loop, no output stores"), (4, "no output stores"), (1 | 2, "hot operands (zero page + one weight line)"),
       (1 | 2 | 4, "hot operands, no output stores"), (64, "no MFMAs"), (32, "no fragment ds_reads"), (128 | 256, "no LDS-DMA in the main loop"),
       (32 | 64 | 128 | 256, "barriers only in the main loop"), (32 | 64 | 128 | 256 | 4, "barriers only, no output stores")]
SHAPES = [(64, 64, 512, 1024), (64, 64, 256, 512), (128, 128, 256, 512)]
with torch.no_grad():
    for cin, cout, H, W in SHAPES:
        conv = nn.Conv2d(cin, cout, 3, padding=0).to(DEV)
        x = eng.pack(torch.randn(1, cin, H, W, device=DEV))
        key = (cin, cout, 3, 1, 0)
        print("== %d -> %d 3x3 @%dx%d, tile 89 (256 px x 64), %d tiles, in %.0f MB + raw fp32 out %.0f MB" % (
            cin, cout, W, H, H * W // 256 * (cout // 64), H * W * cin * 2 / 1e6, H * W * cout * 4 / 1e6), flush=True)
        for ab, what in ABL:
            eng.tile_override[key] = (89, 1, 0)
            eng.ablate = ab
            run = lambda: eng.conv(x, conv, L.PAD_REFLECT, 1, L.OUT_RAW_F32_NHWC, want_stats=True)
            for _ in range(2):
                run()
            ts = []
            for _ in range(9):
                thrash.zero_()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); run(); e1.record(); e1.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            print("   ablate %5d  %7.1f us   %s" % (ab, sorted(ts)[4], what), flush=True)
        eng.ablate = 0
