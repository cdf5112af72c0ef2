#!/usr/bin/env python3
"""Launch FlowNetC's correlation (v2v_correlation_nhwc, the geometry of the 512x256 frame-pair batch: N = 2, C = 256, 32 x 64)
`reps` times, cold cache between launches -- target of rocprofv3 --kernel-trace / --pmc passes (scripts/gpu_r3.sh corrpmc).
    python scripts/corr_run.py [--reps 20] [--precision bf16|fp32]"""
import argparse
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vid2vid_amd import lib as L
from vid2vid_amd.lib import lib, check
from vid2vid_amd.engine import Engine, _ptr, _stream

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--precision", default="bf16")
a = ap.parse_args()
eng = Engine("cuda:0", L.BF16 if a.precision == "bf16" else L.F32)
B, C, H, W = 2, 256, 32, 64
f1, f2 = eng.pack(torch.randn(B, C, H, W, device="cuda:0")), eng.pack(torch.randn(B, C, H, W, device="cuda:0"))
out = torch.zeros((B, H, W, 480), dtype=eng.tdtype, device="cuda:0")
thrash = torch.empty(96 << 20, dtype=torch.float32, device="cuda:0")
for _ in range(a.reps):
    thrash.zero_()
    check(lib.v2v_correlation_nhwc(_ptr(f1.t), _ptr(f2.t), _ptr(out), B, C, H, W, f1.Cs, 480, 32, 20, 2, 0.1, eng.dtype, _stream()), "corr")
torch.cuda.synchronize()
esz = 2 if a.precision == "bf16" else 4
print("algorithmic: %.2f GFLOP (2 N H W 441 C), %.2f MB (both feature maps + 441 output channels)"
      % (2.0 * B * H * W * 441 * C / 1e9, (2 * B * H * W * C * esz + B * H * W * 441 * esz) / 1e6))
