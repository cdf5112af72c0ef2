#!/usr/bin/env python3
"""Micro-benchmark of the ResnetBlock 3x3 convolutions (the dominant layers of the 512x256 frame): round-1 ping-pong tiles
(50-57) vs the second schedule (70-75: LDS-DMA issued between the MFMAs), alone, split-K, and as PAIRED launches
(v2v_conv2d_pair: two convolutions of the same shape in one launch).  HIP-event timing, cold weights (384 MB memset
between launches) and warm, interleaved rounds (median over rounds).

    python scripts/pp2_bench.py [bf16|fp32] > gpurun_out/pp2_bench.txt
"""
import sys
import os
import ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn
from vid2vid_amd import lib as L
from vid2vid_amd.lib import lib
from vid2vid_amd.engine import Engine

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
DEV = "cuda:0"
eng = Engine(DEV, L.BF16 if prec == "bf16" else L.F32)
THRASH = torch.empty(96 << 20, dtype=torch.float32, device=DEV)
ROUNDS = 7


def timed(fn, cold):
    if cold:
        THRASH.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3


def bench(variants, cold):
    """variants: {name: fn}.  Interleaved rounds; returns {name: median us}."""
    for fn in variants.values():
        for _ in range(2):
            fn()
    torch.cuda.synchronize()
    ts = {k: [] for k in variants}
    for _ in range(ROUNDS):
        for k, fn in variants.items():
            ts[k].append(timed(fn, cold))
    return {k: sorted(v)[len(v) // 2] for k, v in ts.items()}


for cin, cout, H, W in ((1024, 1024, 32, 64), (512, 512, 32, 64), (1024, 1024, 64, 64)):
    convs = [nn.Conv2d(cin, cout, 3, padding=0).to(DEV) for _ in range(2)]
    norms = [nn.BatchNorm2d(cout).to(DEV) for _ in range(2)]
    xs = [eng.pack(torch.randn(1, cin, H, W, device=DEV)) for _ in range(2)]
    flops = 2.0 * H * W * cout * cin * 9
    key = (cin, cout, 3, 1, 0)
    singles, pairs = {}, {}

    def single(tile, S):
        def fn():
            eng.tile_override[key] = (tile, S, 0)
            ss = eng.scratch("scale_shift", 4 * cout)
            eng.conv(xs[0], convs[0], L.PAD_REFLECT, 1, L.OUT_RAW_F32_NHWC, want_stats=True, fin=(norms[0], ss))
        return fn

    def pair(tile, S):
        def fn():
            eng.pair_override = (tile, S)
            ssa = eng.scratch("scale_shift", 4 * cout)
            with eng.scratch_set(1):
                ssb = eng.scratch("scale_shift", 4 * cout)
            eng.conv_pair(xs[0], convs[0], xs[1], convs[1], L.PAD_REFLECT, 1, ((norms[0], ssa), (norms[1], ssb)), ("a", "b"))
        return fn

    ncc = xs[0].Cs // (64 if prec == "bf16" else 32)
    for tile, S in ((54, 1), (54, 2), (70, 1),
                    (80, 1), (80, 2), (82, 1), (83, 1), (84, 1), (84, 2), (85, 1), (86, 1), (86, 2), (87, 1)):
        if W % 64 and tile in (51, 57, 73, 74):
            pass
        if 2 * S <= ncc:
            singles["t%d S%d" % (tile, S)] = single(tile, S)
    for tile, S in ((80, 1), (82, 1), (83, 1), (84, 1), (85, 1), (86, 1), (87, 1), (86, 2)):
        if 2 * S <= ncc:
            pairs["pair t%d S%d" % (tile, S)] = pair(tile, S)
    for cold in (True, False):
        try:
            r1 = bench(singles, cold)
            r2 = bench(pairs, cold)
        except RuntimeError as ex:
            print("error:", ex)
            continue
        print("== %d->%d 3x3 @%dx%d %s  %s  (%.2f GFLOP per conv)" % (cin, cout, W, H, prec, "cold" if cold else "warm", flops / 1e9))
        for k, us in r1.items():
            print("   %-14s %7.1f us  %7.0f TFLOP/s" % (k, us, flops / us / 1e6))
        for k, us in r2.items():
            print("   %-14s %7.1f us  %7.0f TFLOP/s (two convs)" % (k, us, 2 * flops / us / 1e6))
        sys.stdout.flush()
