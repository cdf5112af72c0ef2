#!/usr/bin/env python3
"""Where does a step of the second-schedule ping-pong 3x3 kernel go?  Tile ids 79 / 78 are instrumented copies of 70 / 71
(csrc/conv3x3_pp2_kernel.h, ABL = 1); each ablation removes one ingredient of the main loop (results are wrong, only the
time is meaningful).  1024 -> 1024 3x3 at 32x64 pixels: 128 (tile 79) / 64 (tile 78) workgroups, one per CU.

    python scripts/pp2_ablate.py > gpurun_out/pp2_ablate.txt
"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn
from vid2vid_amd import lib as L
from vid2vid_amd.engine import Engine

TILES = [int(t) for t in sys.argv[1:]] or [89, 88, 79]
DEV = "cuda:0"
eng = Engine(DEV, L.BF16)
cin = cout = 1024
H, W = 32, 64
conv = nn.Conv2d(cin, cout, 3, padding=0).to(DEV)
norm = nn.BatchNorm2d(cout).to(DEV)
x = eng.pack(torch.randn(1, cin, H, W, device=DEV))
flops = 2.0 * H * W * cout * cin * 9
key = (cin, cout, 3, 1, 0)
ABL = [(0, "full kernel"), (1 | 2, "hot operands (zero page + one weight line)"), (4, "no output stores"), (1 | 2 | 4, "hot operands, no stores"),
       (32, "no ds_reads"), (64, "no MFMAs"), (32 | 64, "no ds_reads, no MFMAs (DMA + waits + barriers)"),
       (128, "no LDS-DMA"), (128 | 256, "no LDS-DMA, no vmcnt wait"), (128 | 256 | 32, "MFMAs + barriers only"),
       (128 | 256 | 64, "ds_reads + barriers only"), (128 | 256 | 32 | 64, "barriers only"), (256, "no vmcnt wait (DMA never awaited)"),
       (1 | 2 | 256, "hot operands, no vmcnt wait")]


def run(tile, S, ab):
    eng.tile_override[key] = (tile, S, 0)
    eng.ablate = ab
    ss = eng.scratch("scale_shift", 4 * cout)
    eng.conv(x, conv, L.PAD_REFLECT, 1, L.OUT_RAW_F32_NHWC, want_stats=True, fin=(norm, ss))


with torch.no_grad():
    # launch + event overhead of an (almost) empty kernel, for reference
    for tile in TILES:
        print("== tile %d (%s, %s), 1024->1024 3x3 @64x32 bf16, warm, median of 9" % (
            tile, "256 px x 64, wave tile 64x32" if tile in (79, 89) else "256 px x 128, wave tile 64x64",
            "single-phase pp3" if tile >= 80 else "ping-pong pp2"))
        for S in (1, 2):
            for ab, what in ABL:
                for _ in range(3):
                    run(tile, S, ab)
                torch.cuda.synchronize()
                ts = []
                for _ in range(9):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    run(tile, S, ab)
                    e1.record()
                    e1.synchronize()
                    ts.append(e0.elapsed_time(e1) * 1e3)
                us = sorted(ts)[4]
                steps = 16 * 9 // S
                print("   S=%d ablate %3d  %7.1f us  (%5.0f ns/step)  %s" % (S, ab, us, us * 1e3 / steps, what), flush=True)
    eng.ablate = 0
