#!/usr/bin/env python3
"""v2v_conv_wgrad: bf16-MFMA kernel vs the exact-fp32 kernel on the same bf16-representable data (self-consistency
beside the torch-autograd parity tests), and timings on the training step's heaviest layer shapes.
    python scripts/wgrad_bench.py            (V2V_WGRAD_BF16=legacy python ... times the round-1 v1 bf16 path)"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vid2vid_amd import lib as L
from vid2vid_amd.lib import lib, WgradDesc, check

dev = "cuda:0"
zero = torch.zeros(256, dtype=torch.uint8, device=dev)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def run(dy, x, KH, stride, pad, pad_mode, dtype, reps=0):
    """dy [N,OH,OW,R], x [N,H,W,Cc] (NHWC, given dtype) -> grad [R][Cc][KH][KH] fp32 (+ median ms)"""
    N, OH, OW, R = dy.shape
    _, H, W, Cc = x.shape
    d = WgradDesc()
    d.p, d.q = dy.data_ptr(), x.data_ptr()
    d.N, d.OH, d.OW, d.QH, d.QW = N, OH, OW, H, W
    d.rows, d.cols, d.p_stride, d.q_stride = R, Cc, dy.stride(2), x.stride(2)
    d.KH = d.KW = KH
    d.stride, d.pad, d.pad_mode = stride, pad, pad_mode
    d.dtype, d.accumulate = dtype, 0
    grad = torch.zeros(R, Cc, KH, KH, dtype=torch.float32, device=dev)
    d.grad = grad.data_ptr()
    d.zero_page = zero.data_ptr()
    nbytes = lib.v2v_conv_wgrad_workspace(C.byref(d))
    assert nbytes > 0, nbytes
    ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=dev)
    d.workspace = ws.data_ptr()
    check(lib.v2v_conv_wgrad(C.byref(d), st), "wgrad")
    ms = None
    if reps:
        e0 = [torch.cuda.Event(enable_timing=True) for _ in range(reps)]
        e1 = [torch.cuda.Event(enable_timing=True) for _ in range(reps)]
        for r in range(reps):
            e0[r].record(); lib.v2v_conv_wgrad(C.byref(d), st); e1[r].record()
        torch.cuda.synchronize()
        ms = sorted(a.elapsed_time(b) for a, b in zip(e0, e1))[reps // 2]
    torch.cuda.synchronize()
    return grad, ms


CASES = [  # R(cout) Cc(cin) K stride pad mode  N  OH  OW   (x is OH*stride.. sized)
    (1024, 1024, 3, 1, 1, L.PAD_REFLECT, 1, 32, 64),     # ResnetBlock 1024 @512x256: 36 layers / frame
    (512, 512, 3, 1, 1, L.PAD_REFLECT, 1, 32, 64),       # fg tower
    (256, 128, 3, 2, 1, L.PAD_ZERO, 1, 128, 256),
    (128, 108, 7, 1, 3, L.PAD_REFLECT, 1, 256, 512),      # stem (cin stride 112)
    (64, 39, 4, 2, 2, L.PAD_ZERO, 2, 129, 257),            # D first layer
    (200, 136, 3, 1, 1, L.PAD_ZERO, 1, 12, 20),
    (3, 128, 7, 1, 3, L.PAD_REFLECT, 1, 64, 64),           # head: 3 rows
    (72, 64, 3, 1, 1, L.PAD_ZERO, 2, 17, 23),
]
torch.manual_seed(0)
worst = 0.0
os.environ["V2V_WGRAD3"] = "0"          # first table: the GEMM-view kernel + reduce on every shape
for (R, Cc, K, s, p, mode, N, OH, OW) in CASES:
    H, W = (OH - 1) * s + K - 2 * p, (OW - 1) * s + K - 2 * p
    Rs, Cs = (R + 7) // 8 * 8, (Cc + 7) // 8 * 8
    dy = torch.zeros(N, OH, OW, Rs, device=dev); dy[..., :R] = torch.randn(N, OH, OW, R, device=dev)
    x = torch.zeros(N, H, W, Cs, device=dev); x[..., :Cc] = torch.randn(N, H, W, Cc, device=dev)
    dyb, xb = dy.bfloat16(), x.bfloat16()
    ref, ms32 = run(dyb.float(), xb.float(), K, s, p, mode, L.F32, reps=5)
    got, ms16 = run(dyb, xb, K, s, p, mode, L.BF16, reps=11)
    rms = ref.pow(2).mean().sqrt().item()
    err = (got - ref).abs().max().item() / (rms + 1e-12)
    worst = max(worst, err)
    flops = 2.0 * N * OH * OW * R * Cc * K * K
    print("wgrad R=%4d C=%4d k%d s%d %dx%dx%d: bf16 %.3f ms (%.0f TFLOP/s)  fp32 %.3f ms (%.0f TFLOP/s)  max|diff|/rms %.2e"
          % (R, Cc, K, s, N, OH, OW, ms16, flops / ms16 / 1e9, ms32, flops / ms32 / 1e9, err))
print("worst", worst)
assert worst < 2e-3, "bf16-MFMA wgrad disagrees with the exact-fp32 kernel on identical (bf16-representable) operands"

# ---- round 6: the nine-tap 3x3 kernel (conv_wgrad3x3_bf16_kernel) beside the GEMM-view kernel + reduce, same operands ----
NINE = [(1024, 1024, 1, 32, 64, (1, 2)), (512, 512, 1, 32, 64, (2, 4, 8)), (1024, 1024, 1, 64, 64, (1, 2)), (256, 256, 1, 64, 128, (4, 8)),
        (128, 128, 1, 256, 512, (8,)), (64, 64, 1, 256, 512, (8,))]
for (R, Cc, N, OH, OW, splits) in NINE:
    dy = torch.randn(N, OH, OW, R, device=dev).bfloat16()
    x = torch.randn(N, OH, OW, Cc, device=dev).bfloat16()
    flops = 2.0 * N * OH * OW * R * Cc * 9
    os.environ["V2V_WGRAD3"] = "0"
    ref, ms0 = run(dy, x, 3, 1, 1, L.PAD_REFLECT, L.BF16, reps=11)
    os.environ["V2V_WGRAD3"] = "1"
    line = "wgrad 3x3 R=%4d C=%4d %dx%dx%d: GEMM view + reduce %.3f ms (%.0f TFLOP/s)" % (R, Cc, N, OH, OW, ms0, flops / ms0 / 1e9)
    for sp in splits:
        os.environ["V2V_WGRAD3_SPLITS"] = str(sp)
        got, ms = run(dy, x, 3, 1, 1, L.PAD_REFLECT, L.BF16, reps=11)
        err = (got - ref).abs().max().item() / (ref.pow(2).mean().sqrt().item() + 1e-12)
        line += " | nine-tap x%d splits %.3f ms (%.0f TFLOP/s, diff %.1e)" % (sp, ms, flops / ms / 1e9, err)
    os.environ.pop("V2V_WGRAD3_SPLITS")
    print(line)

# ---- round 6: the kernel-row 7x7 kernel (conv_wgrad_krow_bf16_kernel + its reduce) beside the GEMM view, same operands ----
KROW = [(128, 108, 1, 256, 512, (0, 4, 8, 16)), (64, 108, 1, 256, 512, (0, 8, 16)), (128, 6, 1, 256, 512, (0, 8)), (3, 128, 1, 256, 512, (0, 8)),
        (64, 108, 1, 512, 1024, (0, 16)), (32, 108, 1, 1024, 2048, (0, 16, 32)), (3, 32, 1, 1024, 2048, (0, 16)), (32, 6, 1, 1024, 2048, (0, 16))]
if os.environ.get("WGRAD_BENCH_KROW", "1") != "0":
    for (R, Cc, N, OH, OW, splits) in KROW:
        Rs, Cs = (R + 7) // 8 * 8, (Cc + 7) // 8 * 8
        dy = torch.zeros(N, OH, OW, Rs, device=dev); dy[..., :R] = torch.randn(N, OH, OW, R, device=dev)
        x = torch.zeros(N, OH, OW, Cs, device=dev); x[..., :Cc] = torch.randn(N, OH, OW, Cc, device=dev)
        dy, x = dy.bfloat16(), x.bfloat16()
        flops = 2.0 * N * OH * OW * R * Cc * 49
        os.environ["V2V_WGRAD_KROW"] = "0"
        ref, ms0 = run(dy, x, 7, 1, 3, L.PAD_REFLECT, L.BF16, reps=7)
        os.environ["V2V_WGRAD_KROW"] = "1"
        line = "wgrad 7x7 R=%4d C=%4d %dx%dx%d: GEMM view + reduce %.3f ms (%.0f TFLOP/s)" % (R, Cc, N, OH, OW, ms0, flops / ms0 / 1e9)
        for sp in splits:
            if sp:
                os.environ["V2V_WGRAD_KROW_SPLITS"] = str(sp)
            got, ms = run(dy, x, 7, 1, 3, L.PAD_REFLECT, L.BF16, reps=7)
            err = (got - ref).abs().max().item() / (ref.pow(2).mean().sqrt().item() + 1e-12)
            line += " | kernel row %s %.3f ms (%.0f TFLOP/s, diff %.1e)" % ("x%d" % sp if sp else "auto", ms, flops / ms / 1e9, err)
            os.environ.pop("V2V_WGRAD_KROW_SPLITS", None)
        print(line)

# ---- the same kernel with three taps per workgroup on the 3x3 layers the nine-tap kernel leaves to the GEMM view ----
KROW3 = [(128, 128, 1, 256, 512), (64, 64, 1, 256, 512), (64, 64, 1, 512, 1024), (32, 32, 1, 1024, 2048), (64, 32, 1, 1024, 2048), (256, 256, 1, 64, 128)]
if os.environ.get("WGRAD_BENCH_KROW", "1") != "0":
    for (R, Cc, N, OH, OW) in KROW3:
        dy = torch.randn(N, OH, OW, R, device=dev).bfloat16()
        x = torch.randn(N, OH, OW, Cc, device=dev).bfloat16()
        flops = 2.0 * N * OH * OW * R * Cc * 9
        os.environ["V2V_WGRAD_KROW3"] = "0"
        ref, ms0 = run(dy, x, 3, 1, 1, L.PAD_REFLECT, L.BF16, reps=7)
        os.environ["V2V_WGRAD_KROW3"] = "1"
        line = "wgrad 3x3 R=%4d C=%4d %dx%dx%d: as before %.3f ms (%.0f TFLOP/s)" % (R, Cc, N, OH, OW, ms0, flops / ms0 / 1e9)
        for sp in (0, 16, 32, 64):
            if sp:
                os.environ["V2V_WGRAD_KROW_SPLITS"] = str(sp)
            got, ms = run(dy, x, 3, 1, 1, L.PAD_REFLECT, L.BF16, reps=7)
            err = (got - ref).abs().max().item() / (ref.pow(2).mean().sqrt().item() + 1e-12)
            line += " | kernel row %s %.3f ms (%.0f TFLOP/s, diff %.1e)" % ("x%d" % sp if sp else "auto", ms, flops / ms / 1e9, err)
            os.environ.pop("V2V_WGRAD_KROW_SPLITS", None)
        print(line)

# ---- the strided instantiations (4x4 / stride 2 and 1 of the discriminators, 3x3 / stride 2 down layers) beside the GEMM view ----
KROWS = [(64, 39, 4, 2, 2, 2, 129, 257), (128, 64, 4, 2, 2, 2, 65, 129), (256, 128, 4, 2, 2, 2, 33, 65), (512, 256, 4, 1, 2, 2, 34, 66), (1, 512, 4, 1, 2, 2, 35, 67),
         (256, 128, 3, 2, 1, 1, 128, 256), (512, 256, 3, 2, 1, 1, 64, 128), (1024, 512, 3, 2, 1, 1, 32, 64), (64, 39, 4, 2, 2, 2, 513, 1025), (128, 64, 4, 2, 2, 2, 257, 513)]
if os.environ.get("WGRAD_BENCH_KROW", "1") != "0":
    for (R, Cc, K, s, p, N, OH, OW) in KROWS:
        H, W = (OH - 1) * s + K - 2 * p, (OW - 1) * s + K - 2 * p
        Rs, Cs = (R + 7) // 8 * 8, (Cc + 7) // 8 * 8
        dy = torch.zeros(N, OH, OW, Rs, device=dev); dy[..., :R] = torch.randn(N, OH, OW, R, device=dev)
        x = torch.zeros(N, H, W, Cs, device=dev); x[..., :Cc] = torch.randn(N, H, W, Cc, device=dev)
        dy, x = dy.bfloat16(), x.bfloat16()
        flops = 2.0 * N * OH * OW * R * Cc * K * K
        os.environ["V2V_WGRAD_KROW_S"] = "0"
        ref, ms0 = run(dy, x, K, s, p, L.PAD_ZERO, L.BF16, reps=7)
        os.environ["V2V_WGRAD_KROW_S"] = "1"
        line = "wgrad %dx%d/s%d R=%4d C=%4d %dx%dx%d: GEMM view + reduce %.3f ms (%.0f TFLOP/s)" % (K, K, s, R, Cc, N, OH, OW, ms0, flops / ms0 / 1e9)
        for sp in (0, 4, 16, 48):
            if sp:
                os.environ["V2V_WGRAD_KROW_SPLITS"] = str(sp)
            got, ms = run(dy, x, K, s, p, L.PAD_ZERO, L.BF16, reps=7)
            err = (got - ref).abs().max().item() / (ref.pow(2).mean().sqrt().item() + 1e-12)
            line += " | kernel row %s %.3f ms (%.0f TFLOP/s, diff %.1e)" % ("x%d" % sp if sp else "auto", ms, flops / ms / 1e9, err)
            os.environ.pop("V2V_WGRAD_KROW_SPLITS", None)
        print(line)
