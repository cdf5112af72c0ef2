#!/usr/bin/env python3
"""Launch v2v_conv_wgrad (bf16, 3x3 / stride 1 / pad 1) on one layer shape `reps` times -- the target of the rocprofv3 counter
passes of scripts/gpu_r6.sh wgradpmc.      python scripts/wgrad3_run.py R C H W [reps] [accumulate]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vid2vid_amd import lib as L
from vid2vid_amd.lib import lib, WgradDesc, check

R, Cc, H, W = [int(v) for v in sys.argv[1:5]]
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 12
acc = int(sys.argv[6]) if len(sys.argv) > 6 else 1
dev = "cuda:0"
dy = torch.randn(1, H, W, R, device=dev).bfloat16()
x = torch.randn(1, H, W, Cc, device=dev).bfloat16()
zero = torch.zeros(256, dtype=torch.uint8, device=dev)
grad = torch.zeros(R, Cc, 3, 3, device=dev)
d = WgradDesc()
d.p, d.q = dy.data_ptr(), x.data_ptr()
d.N, d.OH, d.OW, d.QH, d.QW = 1, H, W, H, W
d.rows, d.cols, d.p_stride, d.q_stride = R, Cc, R, Cc
d.KH = d.KW = 3
d.stride, d.pad, d.pad_mode = 1, 1, L.PAD_REFLECT
d.dtype, d.accumulate = L.BF16, acc
d.grad, d.zero_page = grad.data_ptr(), zero.data_ptr()
nbytes = lib.v2v_conv_wgrad_workspace(C.byref(d))
ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=dev)
d.workspace = ws.data_ptr()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
thrash = torch.empty(96 << 20, dtype=torch.float32, device=dev)
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
for a, b in ev:
    thrash.zero_()
    a.record(); check(lib.v2v_conv_wgrad(C.byref(d), st), "wgrad"); b.record()
torch.cuda.synchronize()
ms = sorted(a.elapsed_time(b) for a, b in ev)[reps // 2]
print("wgrad 3x3 %d -> %d at %dx%d accumulate=%d: %.1f us cold (%.0f TFLOP/s)" % (Cc, R, W, H, acc, ms * 1e3, 2.0 * H * W * R * Cc * 9 / ms / 1e9))
