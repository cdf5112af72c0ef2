#!/usr/bin/env python3
"""7x7 heads of the 512x256 frame (128 -> 3 tanh, 128 -> 2+1 merged, 64 -> 3) on the halo-patch kernel (tile 60): HIP-event
timing, cold and warm.   python scripts/head_bench.py > gpurun_out/head_bench.txt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn
from vid2vid_amd import lib as L
from vid2vid_amd.engine import Engine
DEV = "cuda:0"
eng = Engine(DEV, L.BF16)
THRASH = torch.empty(96 << 20, dtype=torch.float32, device=DEV)
H, W = 256, 512


def timed(fn, cold, rounds=11):
    for _ in range(2):
        fn()
    ts = []
    for _ in range(rounds):
        if cold:
            THRASH.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[len(ts) // 2]


with torch.no_grad():
    for cin, cout in ((128, 3), (128, 2), (64, 3)):
        x = eng.pack(torch.randn(1, cin, H, W, device=DEV))
        seq = nn.Sequential(nn.ReflectionPad2d(3), nn.Conv2d(cin, cout, 7), nn.Tanh()).to(DEV)
        eng.tile_override[(cin, cout, 7, 1, 0)] = (60, 1, 0)
        fn = lambda: eng.run_sequential(seq, x, head_nchw=True, name="head")
        flops = 2.0 * H * W * cout * cin * 49
        print("7x7 head %3d -> %d @%dx%d  cold %6.1f us  warm %6.1f us   (%.1f GFLOP)" % (cin, cout, W, H, timed(fn, True), timed(fn, False), flops / 1e9))
