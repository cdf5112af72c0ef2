#!/usr/bin/env python3
"""label2city stems (108 -> 128 and 108 -> 64, 7x7, 512x256): dense convolution on the encoded one-hot tensor vs the
weight gather-sum on the label maps (csrc/onehot_stem.hip), with 32- and 64-channel slices per workgroup.  HIP-event timing, cold (384 MB memset between launches) and warm.

    python scripts/stem_bench.py [bf16|fp32] > gpurun_out/stem_bench.txt
"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn
from vid2vid_amd import lib as L
from vid2vid_amd.engine import Engine

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
DEV = "cuda:0"
eng = Engine(DEV, L.BF16 if prec == "bf16" else L.F32)
eng.autotune = True
THRASH = torch.empty(96 << 20, dtype=torch.float32, device=DEV)
T, nc, H, W = 3, 35, 256, 512
torch.manual_seed(0)


def blocky(n, bh, bw, dtype):
    t = torch.randint(0, n, (T, H // bh + 1, W // bw + 1), device=DEV)
    return t.repeat_interleave(bh, 1).repeat_interleave(bw, 2)[:, :H, :W].to(dtype).contiguous()


def timed(fn, cold, rounds=9):
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


for name, labels, inst in (("uniform-random 1x1 px labels (worst case: every lane another row)", blocky(nc, 1, 1, torch.uint8), blocky(50, 1, 1, torch.int32)),
                           ("blocky 16x24 px segments", blocky(nc, 16, 24, torch.uint8), blocky(50, 32, 48, torch.int32)),
                           ("one segment, no edges", torch.zeros(T, H, W, dtype=torch.uint8, device=DEV), torch.zeros(T, H, W, dtype=torch.int32, device=DEV))):
    print("== %s, %s" % (name, prec))
    x, _ = eng.encode_labels(labels, inst, T, H, W, nc, (), False)
    for cout in (128, 64):
        conv = nn.Conv2d(T * (nc + 1), cout, 7).to(DEV)
        norm = nn.BatchNorm2d(cout).to(DEV)
        flops = 2.0 * H * W * cout * T * (nc + 1) * 49
        with torch.no_grad():
            def dense():
                eng.onehot_stem = False
                ss = eng.scratch("scale_shift", 4 * cout)
                eng.conv(x, conv, L.PAD_REFLECT, 3, L.OUT_RAW_F32_NHWC, want_stats=True, fin=(norm, ss))
            dense()                                     # autotunes the dense layer once
            eng.autotune = False
            res = {"dense": dense}
            for v in (32, 64):
                def oh(v=v):
                    eng.onehot_stem = True
                    eng.onehot_slice = v
                    eng.onehot_conv(x, conv)
                res["gather s%d" % v] = oh
            for cold in (True, False):
                for k, fn in res.items():
                    us = timed(fn, cold)
                    print("   cout %3d %-10s %s %8.1f us   (%.0f dense-equivalent TFLOP/s)" % (cout, k, "cold" if cold else "warm", us, flops / us / 1e6))
            eng.autotune = True
    sys.stdout.flush()
