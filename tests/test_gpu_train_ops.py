"""GPU parity tests of the training-path custom ops (forward AND backward kernels through the C ABI)
against torch CPU autograd of the same op in fp32.

Tolerance: the backward kernels accumulate in exact fp32 (v_mfma_f32_32x32x2_f32 / fp32 VALU); only the
summation order differs from ATen's CPU kernels -> 2e-4 per element with the rms floor of
util.assert_close (north_star bound: 1e-3).  bf16 storage mode is checked at 3e-2 on the gradients
(inputs, saved activations and dY are bf16-rounded; accumulation stays fp32).
"""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from util import assert_close

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _engine(prec="fp32"):
    from vid2vid_amd import lib as L
    from vid2vid_amd.engine import Engine
    return Engine(DEV, L.BF16 if prec == "bf16" else L.F32)


def _pad_mode(mode):
    from vid2vid_amd import lib as L
    return L.PAD_REFLECT if mode == "reflect" else L.PAD_ZERO


CONV_BWD_CASES = [
    # cin, cout, k, stride, pad, mode, H, W, N
    (16, 32, 3, 1, 1, "reflect", 16, 24, 2),
    (64, 64, 3, 1, 1, "reflect", 18, 34, 1),
    (12, 8, 7, 1, 3, "reflect", 20, 28, 1),
    (6, 16, 7, 1, 3, "reflect", 9, 11, 2),      # pad 3 on a 9-wide image: overlapping mirror strips
    (32, 64, 3, 2, 1, "zero", 32, 48, 1),
    (32, 48, 3, 2, 1, "zero", 31, 45, 2),       # odd sizes: parity classes with different grids
    (13, 16, 4, 2, 2, "zero", 33, 47, 1),       # PatchGAN layers (networks.py:685-706)
    (16, 32, 4, 2, 2, "zero", 32, 64, 2),
    (16, 32, 4, 1, 2, "zero", 17, 19, 1),
    (32, 1, 4, 1, 2, "zero", 18, 22, 2),
    (128, 3, 7, 1, 3, "reflect", 24, 40, 1),
    (40, 24, 5, 2, 2, "zero", 30, 42, 1),       # FlowNet2 shapes
    (24, 16, 1, 1, 0, "zero", 13, 29, 1),
    (136, 200, 3, 1, 1, "zero", 12, 20, 1),     # several row / column tiles in the wgrad GEMM
]


def _ref_conv(x, conv, k, stride, pad, mode):
    xp = F.pad(x, (pad,) * 4, mode="reflect") if mode == "reflect" else x
    return F.conv2d(xp, conv.weight, conv.bias, stride=stride, padding=0 if mode == "reflect" else pad)


@pytest.mark.parametrize("layout", ["std", "channels_last"])
@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("case", CONV_BWD_CASES)
def test_conv2d_backward(case, prec, layout, monkeypatch):
    """layout channels_last: the parameters live in optim.FlatBuffers (weights and their gradients stored [cout][KH][KW][cin]):
    the forward / backward-data packings read that layout (v2v_conv_pack_weights korder + 256) and the weight gradient is
    written in it (accumulate + 2: straight into .grad when the input's channel stride equals cin, else by the no-transpose
    reduce)."""
    from vid2vid_amd import lib as L
    from vid2vid_amd import autograd as AG
    cin, cout, k, stride, pad, mode, H, W, N = case
    torch.manual_seed(sum(case[:5]) + H)
    eng = _engine(prec)
    conv = nn.Conv2d(cin, cout, k, stride=stride, padding=0 if mode == "reflect" else pad)
    with torch.no_grad():
        conv.weight.normal_(0, 0.2)
        conv.bias.normal_(0, 0.5)
    x = torch.randn(N, cin, H, W)
    rnd = (lambda t: t.bfloat16().float()) if prec == "bf16" else (lambda t: t.clone())
    # ---- CPU autograd reference (operands rounded like the device copy) ----
    xr = rnd(x).requires_grad_(True)
    cref = nn.Conv2d(cin, cout, k, stride=stride, padding=0 if mode == "reflect" else pad)
    with torch.no_grad():
        cref.weight.copy_(rnd(conv.weight)); cref.bias.copy_(conv.bias)
    yr = F.leaky_relu(_ref_conv(xr, cref, k, stride, pad, mode), 0.2)
    r = rnd(torch.randn_like(yr))
    (yr * r).sum().backward()
    # ---- HIP ----
    conv = conv.to(DEV)
    if layout == "channels_last":
        from vid2vid_amd.optim import FlatBuffers
        monkeypatch.setenv("V2V_WEIGHTS_CL", "1")
        fb = FlatBuffers([conv.weight, conv.bias])
        assert fb.channels_last and (k == 1 or (AG.is_channels_last(conv.weight) and AG.is_channels_last(conv.weight.grad)))
    xg = x.to(DEV).requires_grad_(True)
    xa = eng.pack(xg)
    assert xa.t.requires_grad
    ya = AG.conv_group(eng, xa, conv, _pad_mode(mode), pad, None, L.ACT_LEAKY, 0.2, None, None, False, 1.0, "t")
    y = eng.unpack(ya)
    tol_f = 1e-4 if prec == "fp32" else 1e-2
    assert_close(y.detach().cpu(), yr.detach(), tol_f, "forward " + str(case))
    (y * r.to(DEV)).sum().backward()
    tol = 2e-4 if prec == "fp32" else 3e-2
    assert_close(xg.grad.cpu(), xr.grad, tol, "dX " + str(case))
    assert_close(conv.weight.grad.cpu(), cref.weight.grad, tol, "dW " + str(case))
    assert_close(conv.bias.grad.cpu(), cref.bias.grad, tol, "db " + str(case))
    # gradient accumulation into an existing .grad buffer (kernels add in place)
    y2 = eng.unpack(AG.conv_group(eng, eng.pack(xg), conv, _pad_mode(mode), pad, None, L.ACT_LEAKY, 0.2, None, None,
                                  False, 1.0, "t"))
    (y2 * r.to(DEV)).sum().backward()
    assert_close(conv.weight.grad.cpu(), 2 * cref.weight.grad, tol, "dW accumulate " + str(case))


WGRAD3_CASES = [
    # rows (cout), cols (cin), H, W, N, mode, forced splits (0 = the library's choice), channels-last gradient
    (1024, 1024, 32, 64, 1, "reflect", 0, False),     # the 36 ResnetBlock layers of the 512x256 frame: 256 tiles, unsplit
    (512, 512, 32, 64, 1, "reflect", 0, False),       # foreground tower: 64 tiles x 4 in-order K splits
    (96, 72, 19, 70, 2, "zero", 3, False),            # ragged rows / cols / width (second segment: 6 pixels), splits that cross the image boundary
    (64, 40, 5, 33, 1, "reflect", 1, True),           # channels-last gradient, one partial segment
    (160, 64, 9, 130, 1, "zero", 4, True),            # three segments, split, channels-last
    (64, 64, 2, 16, 3, "reflect", 2, False),          # two-row images (reflect of both neighbours), batch 3
]


@pytest.mark.parametrize("case", WGRAD3_CASES)
def test_wgrad_nine_tap_kernel(case, monkeypatch):
    """conv_wgrad3x3_bf16_kernel (round 6; 3x3 / stride 1 / pad 1, all nine taps per workgroup, the gradient written straight
    into .grad, in-order K splits) against torch's weight gradient of the same bf16-rounded operands and against the GEMM-view
    kernel + reduce it replaces (V2V_WGRAD3=0): overwrite and accumulate, PyTorch layout and channels-last, reflect and zero
    padding, ragged tiles, several images, splits crossing image boundaries; the split tickets re-arm (second launch)."""
    import ctypes as C
    from vid2vid_amd import lib as L
    from vid2vid_amd.lib import lib, WgradDesc, check
    R, Cc, H, W, N, mode, splits, cl = case
    torch.manual_seed(R + Cc + W)
    Rs, Cs = (R + 7) // 8 * 8, (Cc + 7) // 8 * 8
    dy = torch.zeros(N, H, W, Rs, device=DEV); dy[..., :R] = torch.randn(N, H, W, R, device=DEV)
    x = torch.zeros(N, H, W, Cs, device=DEV); x[..., :Cc] = torch.randn(N, H, W, Cc, device=DEV)
    dyb, xb = dy.bfloat16(), x.bfloat16()
    zero = torch.zeros(256, dtype=torch.uint8, device=DEV)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def run(grad, accumulate):
        d = WgradDesc()
        d.p, d.q = dyb.data_ptr(), xb.data_ptr()
        d.N, d.OH, d.OW, d.QH, d.QW = N, H, W, H, W
        d.rows, d.cols, d.p_stride, d.q_stride = R, Cc, Rs, Cs
        d.KH = d.KW = 3
        d.stride, d.pad, d.pad_mode = 1, 1, L.PAD_REFLECT if mode == "reflect" else L.PAD_ZERO
        d.dtype, d.accumulate = L.BF16, (1 if accumulate else 0) + (2 if cl else 0)
        d.grad, d.zero_page = grad.data_ptr(), zero.data_ptr()
        nbytes = lib.v2v_conv_wgrad_workspace(C.byref(d))
        assert nbytes > 0
        ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=DEV)
        d.workspace = ws.data_ptr()
        check(lib.v2v_conv_wgrad(C.byref(d), st), "wgrad")
        torch.cuda.synchronize()

    shape = (R, 3, 3, Cc) if cl else (R, Cc, 3, 3)
    logical = (lambda g: g.permute(0, 3, 1, 2)) if cl else (lambda g: g)
    # torch: dW of F.conv2d over the padded input, fp32 arithmetic on the bf16-rounded operands
    xr = xb[..., :Cc].float().permute(0, 3, 1, 2).contiguous()
    dyr = dyb[..., :R].float().permute(0, 3, 1, 2).contiguous()
    xp = F.pad(xr, (1, 1, 1, 1), mode="reflect") if mode == "reflect" else F.pad(xr, (1, 1, 1, 1))
    ref = torch.nn.grad.conv2d_weight(xp.double(), (R, Cc, 3, 3), dyr.double()).float().cpu()
    if splits:
        monkeypatch.setenv("V2V_WGRAD3_SPLITS", str(splits))
    g_new = torch.full(shape, 7.0, device=DEV)                             # overwrite mode must not read the buffer
    run(g_new, False)
    monkeypatch.setenv("V2V_WGRAD3", "0")
    g_old = torch.zeros(shape, device=DEV)
    run(g_old, False)
    monkeypatch.setenv("V2V_WGRAD3", "1")
    rms = ref.pow(2).mean().sqrt().item()
    e_ref = (logical(g_new).cpu() - ref).abs().max().item() / rms
    e_old = (g_new - g_old).abs().max().item() / rms
    print("nine-tap wgrad %s: vs torch %.2e, vs the GEMM-view kernel %.2e (of the gradient's rms)" % (str(case), e_ref, e_old))
    assert e_ref < 2e-5 and e_old < 2e-5
    base = torch.randn(shape, device=DEV)
    g_acc = base.clone()
    run(g_acc, True)                                                       # accumulate into .grad; also the second use of the tickets
    assert (g_acc - base - g_new).abs().max().item() / rms < 1e-6
    g_again = torch.empty(shape, device=DEV)
    run(g_again, False)
    assert torch.equal(g_again, g_new)                                     # deterministic (in-order splits), tickets re-armed


WGRAD_KROW_CASES = [
    # rows (cout), cols (cin), H, W, N, mode, forced splits (0 = the library's choice), channels-last gradient
    (128, 108, 64, 128, 1, "reflect", 0, False),      # the label stems (108 one-hot channels -> ngf), library's split count
    (32, 108, 20, 70, 2, "reflect", 3, False),        # half-empty row tile (the 2048x1024 scale's stems), ragged width, splits crossing the image boundary
    (3, 32, 24, 64, 1, "reflect", 2, False),          # a head: 3 gradient rows
    (64, 6, 9, 33, 1, "zero", 1, False),              # previous-frame stem (6 channels), zero padding, one partial segment, unsplit
    (96, 72, 8, 130, 1, "zero", 2, True),             # channels-last gradient, three segments, 4-row splits
    (16, 40, 7, 16, 3, "reflect", 5, False),          # images of 7 rows (reflect reaches 3 rows back), batch 3
    # the same kernel with THREE taps per workgroup: 3x3 layers the nine-tap kernel leaves to the GEMM view (forced splits select it at test sizes)
    (96, 72, 19, 70, 2, "zero", 3, False, 3),
    (64, 64, 12, 130, 1, "reflect", 2, True, 3),
    (40, 33, 6, 64, 2, "reflect", 4, False, 3),
]


@pytest.mark.parametrize("case", WGRAD_KROW_CASES)
def test_wgrad_kernel_row_7x7(case, monkeypatch):
    """conv_wgrad_krow_bf16_kernel (round 6; 7x7 / stride 1 / pad 3: the seven taps of one kernel row per workgroup, K splits parked
    in slabs, wgrad_krow_reduce_kernel sums them in split order and scatters into .grad) against torch's weight gradient of the
    same bf16-rounded operands (fp64) and against the GEMM-view kernel it replaces (V2V_WGRAD_KROW=0): overwrite and accumulate,
    both gradient layouts, reflect and zero padding, ragged tiles / widths, batches, splits crossing image boundaries, repeat
    launches bit-identical."""
    import ctypes as C
    from vid2vid_amd import lib as L
    from vid2vid_amd.lib import lib, WgradDesc, check
    R, Cc, H, W, N, mode, splits, cl = case[:8]
    K = case[8] if len(case) > 8 else 7
    pd = K // 2
    torch.manual_seed(R + Cc + W)
    Rs, Cs = (R + 7) // 8 * 8, (Cc + 7) // 8 * 8
    dy = torch.zeros(N, H, W, Rs, device=DEV); dy[..., :R] = torch.randn(N, H, W, R, device=DEV)
    x = torch.zeros(N, H, W, Cs, device=DEV); x[..., :Cc] = torch.randn(N, H, W, Cc, device=DEV)
    dyb, xb = dy.bfloat16(), x.bfloat16()
    zero = torch.zeros(256, dtype=torch.uint8, device=DEV)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def run(grad, accumulate):
        d = WgradDesc()
        d.p, d.q = dyb.data_ptr(), xb.data_ptr()
        d.N, d.OH, d.OW, d.QH, d.QW = N, H, W, H, W
        d.rows, d.cols, d.p_stride, d.q_stride = R, Cc, Rs, Cs
        d.KH = d.KW = K
        d.stride, d.pad, d.pad_mode = 1, pd, L.PAD_REFLECT if mode == "reflect" else L.PAD_ZERO
        d.dtype, d.accumulate = L.BF16, (1 if accumulate else 0) + (2 if cl else 0)
        d.grad, d.zero_page = grad.data_ptr(), zero.data_ptr()
        nbytes = lib.v2v_conv_wgrad_workspace(C.byref(d))
        assert nbytes > 0
        ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=DEV)
        d.workspace = ws.data_ptr()
        check(lib.v2v_conv_wgrad(C.byref(d), st), "wgrad")
        torch.cuda.synchronize()

    shape = (R, K, K, Cc) if cl else (R, Cc, K, K)
    logical = (lambda g: g.permute(0, 3, 1, 2)) if cl else (lambda g: g)
    xr = xb[..., :Cc].float().permute(0, 3, 1, 2).contiguous()
    dyr = dyb[..., :R].float().permute(0, 3, 1, 2).contiguous()
    xp = F.pad(xr, (pd,) * 4, mode="reflect") if mode == "reflect" else F.pad(xr, (pd,) * 4)
    ref = torch.nn.grad.conv2d_weight(xp.double(), (R, Cc, K, K), dyr.double()).float().cpu()
    if splits:
        monkeypatch.setenv("V2V_WGRAD_KROW_SPLITS", str(splits))
    g_new = torch.full(shape, 7.0, device=DEV)                             # overwrite mode must not read the buffer
    run(g_new, False)
    monkeypatch.setenv("V2V_WGRAD_KROW", "0")
    g_old = torch.zeros(shape, device=DEV)
    run(g_old, False)
    monkeypatch.setenv("V2V_WGRAD_KROW", "1")
    rms = ref.pow(2).mean().sqrt().item()
    e_ref = (logical(g_new).cpu() - ref).abs().max().item() / rms
    e_old = (g_new - g_old).abs().max().item() / rms
    print("kernel-row wgrad %s: vs torch %.2e, vs the GEMM-view kernel %.2e (of the gradient's rms)" % (str(case), e_ref, e_old))
    assert e_ref < 2e-5 and e_old < 2e-5
    base = torch.randn(shape, device=DEV)
    g_acc = base.clone()
    run(g_acc, True)
    assert (g_acc - base - g_new).abs().max().item() / rms < 1e-6
    g_again = torch.empty(shape, device=DEV)
    run(g_again, False)
    assert torch.equal(g_again, g_new)


WGRAD_KROW_STRIDED_CASES = [
    # rows, cols, OH, OW (dY grid), QH, QW (X grid), N, K, stride, pad, forced splits, channels-last gradient
    (64, 39, 17, 33, 32, 64, 2, 4, 2, 2, 3, False),      # discriminator layer 1: 4x4 / stride 2 / pad 2 (OH = H/2 + 1), ragged columns
    (128, 64, 9, 70, 16, 138, 1, 4, 2, 2, 2, False),     # 4x4 / stride 2, two dY segments (the second: 6 pixels)
    (96, 128, 10, 35, 9, 34, 1, 4, 1, 2, 2, True),       # the discriminators' stride-1 4x4 layers (OH = H + 1), channels-last gradient
    (1, 72, 6, 20, 5, 19, 3, 4, 1, 2, 4, False),         # the last layer: ONE gradient row
    (80, 64, 8, 40, 16, 80, 2, 3, 2, 1, 3, False),       # generator down layer: 3x3 / stride 2 / pad 1
    (64, 40, 12, 64, 24, 128, 1, 3, 2, 1, 2, False),     # ConvTranspose2d(3x3, stride 2, pad 1, output_padding 1): P = the layer's input, Q = dY (twice the size)
]


@pytest.mark.parametrize("case", WGRAD_KROW_STRIDED_CASES)
def test_wgrad_kernel_row_strided(case, monkeypatch):
    """conv_wgrad_krow_bf16_kernel<2, 4 | 3, 1 | 2>: the strided layers (X row segments staged de-interleaved for stride 2) against
    torch's fp64 weight gradient of the same bf16 operands and against the GEMM-view kernel (V2V_WGRAD_KROW=0); overwrite, accumulate,
    both gradient layouts, one gradient row, splits crossing image boundaries, bit-identical repeats."""
    import ctypes as C
    from vid2vid_amd import lib as L
    from vid2vid_amd.lib import lib, WgradDesc, check
    R, Cc, OH, OW, QH, QW, N, K, stride, pad, splits, cl = case
    torch.manual_seed(R + Cc + OW)
    Rs, Cs = (R + 7) // 8 * 8, (Cc + 7) // 8 * 8
    dy = torch.zeros(N, OH, OW, Rs, device=DEV); dy[..., :R] = torch.randn(N, OH, OW, R, device=DEV)
    x = torch.zeros(N, QH, QW, Cs, device=DEV); x[..., :Cc] = torch.randn(N, QH, QW, Cc, device=DEV)
    dyb, xb = dy.bfloat16(), x.bfloat16()
    zero = torch.zeros(256, dtype=torch.uint8, device=DEV)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def run(grad, accumulate):
        d = WgradDesc()
        d.p, d.q = dyb.data_ptr(), xb.data_ptr()
        d.N, d.OH, d.OW, d.QH, d.QW = N, OH, OW, QH, QW
        d.rows, d.cols, d.p_stride, d.q_stride = R, Cc, Rs, Cs
        d.KH = d.KW = K
        d.stride, d.pad, d.pad_mode = stride, pad, L.PAD_ZERO
        d.dtype, d.accumulate = L.BF16, (1 if accumulate else 0) + (2 if cl else 0)
        d.grad, d.zero_page = grad.data_ptr(), zero.data_ptr()
        nbytes = lib.v2v_conv_wgrad_workspace(C.byref(d))
        assert nbytes > 0
        ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=DEV)
        d.workspace = ws.data_ptr()
        check(lib.v2v_conv_wgrad(C.byref(d), st), "wgrad")
        torch.cuda.synchronize()

    shape = (R, K, K, Cc) if cl else (R, Cc, K, K)
    logical = (lambda g: g.permute(0, 3, 1, 2)) if cl else (lambda g: g)
    xr = xb[..., :Cc].double().permute(0, 3, 1, 2).contiguous().cpu()
    dyr = dyb[..., :R].double().permute(0, 3, 1, 2).contiguous().cpu()
    # dW[r][c][ky][kx] = sum_p dY[p][r] X[p * stride + k - pad][c] (zero outside): the weight gradient of a strided correlation
    xp = F.pad(xr, (pad, pad + stride + K, pad, pad + stride + K))
    ref = torch.zeros(R, Cc, K, K, dtype=torch.float64)
    for ky in range(K):
        for kx in range(K):
            win = xp[:, :, ky:ky + stride * OH:stride, kx:kx + stride * OW:stride][:, :, :OH, :OW]
            ref[:, :, ky, kx] = torch.einsum("nrhw,nchw->rc", dyr, win)
    ref = ref.float()
    monkeypatch.setenv("V2V_WGRAD_KROW_SPLITS", str(splits))
    g_new = torch.full(shape, 7.0, device=DEV)
    run(g_new, False)
    monkeypatch.setenv("V2V_WGRAD_KROW", "0")
    g_old = torch.zeros(shape, device=DEV)
    run(g_old, False)
    monkeypatch.setenv("V2V_WGRAD_KROW", "1")
    rms = ref.pow(2).mean().sqrt().item()
    e_ref = (logical(g_new).cpu() - ref).abs().max().item() / rms
    e_old = (g_new - g_old).abs().max().item() / rms
    print("strided kernel-row wgrad %s: vs torch %.2e, vs the GEMM-view kernel %.2e (of the gradient's rms)" % (str(case), e_ref, e_old))
    assert e_ref < 2e-5 and e_old < 2e-5
    base = torch.randn(shape, device=DEV)
    g_acc = base.clone()
    run(g_acc, True)
    assert (g_acc - base - g_new).abs().max().item() / rms < 1e-6
    g_again = torch.empty(shape, device=DEV)
    run(g_again, False)
    assert torch.equal(g_again, g_new)


CONVT_CASES = [
    # cin, cout, k, pad, out_pad, H, W, N
    (32, 16, 3, 1, 1, 12, 20, 2),      # generator up path (networks.py:176)
    (64, 32, 3, 1, 1, 9, 7, 1),
    (24, 8, 4, 1, 0, 10, 14, 1),       # FlowNet2 deconv (submodules.py:24-31)
]


@pytest.mark.parametrize("layout", ["std", "channels_last"])
@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("case", CONVT_CASES)
def test_conv_transpose2d_backward(case, prec, layout, monkeypatch):
    from vid2vid_amd import lib as L
    from vid2vid_amd import autograd as AG
    cin, cout, k, pad, op, H, W, N = case
    torch.manual_seed(sum(case))
    eng = _engine(prec)
    conv = nn.ConvTranspose2d(cin, cout, k, stride=2, padding=pad, output_padding=op)
    with torch.no_grad():
        conv.weight.normal_(0, 0.2)
        conv.bias.normal_(0, 0.5)
    rnd = (lambda t: t.bfloat16().float()) if prec == "bf16" else (lambda t: t.clone())
    x = torch.randn(N, cin, H, W)
    xr = rnd(x).requires_grad_(True)
    cref = nn.ConvTranspose2d(cin, cout, k, stride=2, padding=pad, output_padding=op)
    with torch.no_grad():
        cref.weight.copy_(rnd(conv.weight)); cref.bias.copy_(conv.bias)
    yr = F.relu(cref(xr))
    r = rnd(torch.randn_like(yr))
    (yr * r).sum().backward()
    conv = conv.to(DEV)
    if layout == "channels_last":                          # weight [cin][cout][k][k] stored [cin][k][k][cout] in the flat buffers
        from vid2vid_amd.optim import FlatBuffers
        monkeypatch.setenv("V2V_WEIGHTS_CL", "1")
        fb = FlatBuffers([conv.weight, conv.bias])
        assert fb.channels_last and AG.is_channels_last(conv.weight.grad)
    xg = x.to(DEV).requires_grad_(True)
    ya = AG.conv_group(eng, eng.pack(xg), conv, L.PAD_ZERO, None, None, L.ACT_RELU, 0.0, None, None, False, 1.0, "t")
    y = eng.unpack(ya)
    assert_close(y.detach().cpu(), yr.detach(), 1e-4 if prec == "fp32" else 1e-2, "forward " + str(case))
    (y * r.to(DEV)).sum().backward()
    tol = 2e-4 if prec == "fp32" else 3e-2
    assert_close(xg.grad.cpu(), xr.grad, tol, "dX " + str(case))
    assert_close(conv.weight.grad.cpu(), cref.weight.grad, tol, "dW " + str(case))
    assert_close(conv.bias.grad.cpu(), cref.bias.grad, tol, "db " + str(case))


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("case", [("conv", 64, 128, 17, 29, 110), ("conv", 128, 64, 16, 32, 111), ("conv", 64, 192, 32, 32, 113),
                                  ("convT", 128, 64, 9, 13, 100), ("convT", 64, 128, 16, 16, 101), ("convT", 192, 64, 8, 32, 103)])
def test_backward_data_on_the_patch_kernels(case, prec):
    """Round 4: backward-data of a stride-2 3x3 Conv2d IS a transposed stride-2 convolution of dY with the role-swapped weights ->
    conv3x3_t2_kernel (korder-2 packing of role 'bwd'); backward-data of a ConvTranspose2d(3x3, s2) IS a stride-2 convolution of dY ->
    conv3x3_s2_kernel (korder 1, role 'bwd').  The tile search of the training step may select them (Engine.tune_backward_data);
    here they are forced and dX / dW / db are checked against CPU autograd, odd sizes included."""
    from vid2vid_amd import lib as L
    from vid2vid_amd import autograd as AG
    kind, cin, cout, H, W, tile = case
    bke = 64 if prec == "bf16" else 32
    torch.manual_seed(cin + cout + tile)
    eng = _engine(prec)
    if kind == "conv":
        conv = nn.Conv2d(cin, cout, 3, stride=2, padding=1)
        cref = nn.Conv2d(cin, cout, 3, stride=2, padding=1)
        eng.bwd_tile_override[(cout, cin, 3, 2, 1)] = (tile, 1, 0)          # dY (cout channels) -> dX (cin channels), transposed
    else:
        conv = nn.ConvTranspose2d(cin, cout, 3, stride=2, padding=1, output_padding=1)
        cref = nn.ConvTranspose2d(cin, cout, 3, stride=2, padding=1, output_padding=1)
        eng.bwd_tile_override[(cout, cin, 3, 2, 0)] = (tile, 1, 0)          # a stride-2 Conv2d of dY
    if cout % bke != 0:
        pytest.skip("dY channel stride must be whole 128-byte chunks")
    with torch.no_grad():
        conv.weight.normal_(0, 0.2); conv.bias.normal_(0, 0.5)
    rnd = (lambda t: t.bfloat16().float()) if prec == "bf16" else (lambda t: t.clone())
    x = torch.randn(2, cin, H, W)
    xr = rnd(x).requires_grad_(True)
    with torch.no_grad():
        cref.weight.copy_(rnd(conv.weight)); cref.bias.copy_(conv.bias)
    yr = cref(xr)
    r = rnd(torch.randn_like(yr))
    (yr * r).sum().backward()
    conv = conv.to(DEV)
    xg = x.to(DEV).requires_grad_(True)
    ya = AG.conv_group(eng, eng.pack(xg), conv, L.PAD_ZERO, None, None, L.ACT_NONE, 0.0, None, None, False, 1.0, "t")
    y = eng.unpack(ya)
    assert_close(y.detach().cpu(), yr.detach(), 1e-4 if prec == "fp32" else 1e-2, "forward " + str(case))
    (y * r.to(DEV)).sum().backward()
    assert any(c.get("kind") == "bwd_data" for c in eng.conv_log)
    tol = 2e-4 if prec == "fp32" else 3e-2
    assert_close(xg.grad.cpu(), xr.grad, tol, "dX " + str(case))
    assert_close(conv.weight.grad.cpu(), cref.weight.grad, tol, "dW " + str(case))


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("case", [(128, 64, 16, 32, "reflect", 82, 1), (64, 128, 17, 29, "reflect", 90, 1), (128, 128, 32, 64, "reflect", 90, 2),
                                  (64, 192, 9, 40, "zero", 80, 1), (192, 64, 32, 32, "zero", 83, 1), (128, 128, 8, 64, "reflect", 86, 1)])
def test_backward_data_of_stride1_3x3_on_the_single_phase_tiles(case, prec):
    """Round 6: backward-data of a 3x3 / stride 1 Conv2d as a CONVOLUTION on the single-phase 3x3 tiles -- the role-swapped parameter
    with flipped taps (PackedConv korder 4, v2v_conv_pack_weights korder 4) and pad 2 - p: behind a ReflectionPad2d a "full"
    convolution onto the (H+2) x (W+2) padded grid (the kernels' patch origin follows v2v_conv_desc.pad), folded by reflect_pad_fold;
    with zero padding the same-size convolution.  Forced through bwd_tile_override; dX / dW against CPU autograd, odd sizes, split-K,
    ragged tiles (the padded grid is never a multiple of the tile)."""
    from vid2vid_amd import lib as L
    from vid2vid_amd import autograd as AG
    cin, cout, H, W, mode, tile, S = case
    bke = 64 if prec == "bf16" else 32
    if cout % bke != 0:
        pytest.skip("dY channel stride must be whole 128-byte chunks")
    torch.manual_seed(cin + cout + tile + H)
    eng = _engine(prec)
    conv = nn.Conv2d(cin, cout, 3, stride=1, padding=0 if mode == "reflect" else 1)
    cref = nn.Conv2d(cin, cout, 3, stride=1, padding=0 if mode == "reflect" else 1)
    eng.bwd_tile_override[(cout, cin, 3, 1, 1)] = (tile, S, 0)                # dY (cout channels) -> dX (cin channels); role-swapped = "transposed" stride 1
    with torch.no_grad():
        conv.weight.normal_(0, 0.2); conv.bias.normal_(0, 0.5)
    rnd = (lambda t: t.bfloat16().float()) if prec == "bf16" else (lambda t: t.clone())
    x = torch.randn(2, cin, H, W)
    xr = rnd(x).requires_grad_(True)
    with torch.no_grad():
        cref.weight.copy_(rnd(conv.weight)); cref.bias.copy_(conv.bias)
    yr = _ref_conv(xr, cref, 3, 1, 1, mode)
    r = rnd(torch.randn_like(yr))
    (yr * r).sum().backward()
    conv = conv.to(DEV)
    xg = x.to(DEV).requires_grad_(True)
    ya = AG.conv_group(eng, eng.pack(xg), conv, _pad_mode(mode), 1, None, L.ACT_NONE, 0.0, None, None, False, 1.0, "t")
    y = eng.unpack(ya)
    assert_close(y.detach().cpu(), yr.detach(), 1e-4 if prec == "fp32" else 1e-2, "forward " + str(case))
    n0 = len(eng.conv_log)
    (y * r.to(DEV)).sum().backward()
    assert any(c.get("kind") == "bwd_data" for c in eng.conv_log[n0:])
    assert any(pc.korder == 4 for pc in eng._packed.values()), "the backward-data operator did not take the convolution form"
    tol = 2e-4 if prec == "fp32" else 3e-2
    assert_close(xg.grad.cpu(), xr.grad, tol, "dX " + str(case))
    assert_close(conv.weight.grad.cpu(), cref.weight.grad, tol, "dW " + str(case))


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("case", [(128, 3, 16, 32, "reflect"), (64, 2, 19, 45, "reflect"), (32, 3, 40, 72, "reflect"), (32, 1, 8, 33, "zero"),
                                  (16, 3, 24, 64, "reflect")])
def test_backward_data_of_the_7x7_heads_on_the_16_byte_pixel_kernel(case, prec):
    """Round 6: backward-data of the generator heads (ngf -> 3 / 2 / 1 channels, 7x7 behind ReflectionPad2d(3)) on conv7x7_c8_kernel
    (tile 61): the output gradient is one 16-byte vector per pixel, the operator a 7x7 convolution with the role-swapped, tap-flipped
    weights (v2v_conv_pack_weights korder 5) and zero padding 6 - p onto the (H+6) x (W+6) padded grid (activation-typed NHWC output),
    folded by reflect_pad_fold; zero padding 3: the same-size convolution.  Forced through bwd_tile_override; dX / dW against CPU
    autograd; ragged tiles, every N-tile count of the kernel (16 ... 128 dX channels)."""
    from vid2vid_amd import lib as L
    from vid2vid_amd import autograd as AG
    cin, cout, H, W, mode = case
    torch.manual_seed(cin + cout + H)
    eng = _engine(prec)
    pad = 0 if mode == "reflect" else 3
    conv = nn.Conv2d(cin, cout, 7, stride=1, padding=pad)
    cref = nn.Conv2d(cin, cout, 7, stride=1, padding=pad)
    eng.bwd_tile_override[(cout, cin, 7, 1, 1)] = (61, 1, 0)
    with torch.no_grad():
        conv.weight.normal_(0, 0.1); conv.bias.normal_(0, 0.5)
    rnd = (lambda t: t.bfloat16().float()) if prec == "bf16" else (lambda t: t.clone())
    x = torch.randn(2, cin, H, W)
    xr = rnd(x).requires_grad_(True)
    with torch.no_grad():
        cref.weight.copy_(rnd(conv.weight)); cref.bias.copy_(conv.bias)
    yr = torch.tanh(_ref_conv(xr, cref, 7, 1, 3, mode))
    r = rnd(torch.randn_like(yr))
    (yr * r).sum().backward()
    conv = conv.to(DEV)
    xg = x.to(DEV).requires_grad_(True)
    y = AG.conv_group(eng, eng.pack(xg), conv, _pad_mode(mode), 3, None, L.ACT_TANH, 0.0, None, None, True, 1.0, "head")
    assert_close(y.detach().cpu(), yr.detach(), 1e-4 if prec == "fp32" else 2e-2, "forward " + str(case))
    n0 = len(eng.conv_log)
    (y * r.to(DEV)).sum().backward()
    assert any(c.get("kind") == "bwd_data" for c in eng.conv_log[n0:])
    assert any(pc.korder == 5 for pc in eng._packed.values()), "the backward-data operator did not take the 16-byte-pixel kernel"
    tol = 2e-4 if prec == "fp32" else 3e-2
    assert_close(xg.grad.cpu(), xr.grad, tol, "dX " + str(case))
    assert_close(conv.weight.grad.cpu(), cref.weight.grad, tol, "dW " + str(case))


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("norm_kind", ["batch", "instance"])
@pytest.mark.parametrize("act", ["relu", "leaky", "none"])
def test_conv_norm_act_residual_backward(norm_kind, act, prec):
    """ResnetBlock-style group: reflect conv + training-mode norm + activation + two residual adds."""
    from vid2vid_amd import lib as L
    from vid2vid_amd import autograd as AG
    from vid2vid_amd.engine import Act
    torch.manual_seed(11)
    eng = _engine(prec)
    N, C, H, W = (2 if norm_kind == "batch" else 1), 24, 14, 18
    conv = nn.Conv2d(C, C, 3)
    norm = nn.BatchNorm2d(C, affine=True) if norm_kind == "batch" else nn.InstanceNorm2d(C, affine=False)
    with torch.no_grad():
        conv.weight.normal_(0, 0.2); conv.bias.normal_(0, 0.3)
        if norm_kind == "batch":
            norm.weight.normal_(1, 0.2); norm.bias.normal_(0, 0.3)
    rnd = (lambda t: t.bfloat16().float()) if prec == "bf16" else (lambda t: t.clone())
    x, a0, a1 = torch.randn(N, C, H, W), torch.randn(N, C, H, W), torch.randn(N, C, H, W)
    xr, a0r, a1r = [rnd(t).requires_grad_(True) for t in (x, a0, a1)]
    cref = nn.Conv2d(C, C, 3)
    with torch.no_grad():
        cref.weight.copy_(rnd(conv.weight)); cref.bias.copy_(conv.bias)
    raw = cref(F.pad(xr, (1,) * 4, mode="reflect"))
    if norm_kind == "batch":
        gam = norm.weight.detach().clone().requires_grad_(True)
        bet = norm.bias.detach().clone().requires_grad_(True)
        h = F.batch_norm(raw, None, None, gam, bet, True, 0.1, 1e-5)
    else:
        h = F.instance_norm(raw, eps=1e-5)
    h = {"relu": F.relu, "leaky": lambda t: F.leaky_relu(t, 0.2), "none": lambda t: t}[act](h)
    yr = h + a0r + a1r
    r = rnd(torch.randn_like(yr))
    (yr * r).sum().backward()

    conv, norm = conv.to(DEV), norm.to(DEV)
    xg, a0g, a1g = [t.to(DEV).requires_grad_(True) for t in (x, a0, a1)]
    code = {"relu": (L.ACT_RELU, 0.0), "leaky": (L.ACT_LEAKY, 0.2), "none": (L.ACT_NONE, 0.0)}[act]
    ya = AG.conv_group(eng, eng.pack(xg), conv, L.PAD_REFLECT, 1, norm, code[0], code[1], eng.pack(a0g), eng.pack(a1g),
                       False, 1.0, "t")
    y = eng.unpack(ya)
    assert_close(y.detach().cpu(), yr.detach(), 2e-4 if prec == "fp32" else 2e-2, "forward")
    (y * r.to(DEV)).sum().backward()
    tol = 3e-4 if prec == "fp32" else 4e-2
    assert_close(xg.grad.cpu(), xr.grad, tol, "dX")
    assert_close(a0g.grad.cpu(), a0r.grad, tol, "d add0")
    assert_close(a1g.grad.cpu(), a1r.grad, tol, "d add1")
    assert_close(conv.weight.grad.cpu(), cref.weight.grad, tol, "dW")
    if norm_kind == "batch":
        assert_close(norm.weight.grad.cpu(), gam.grad, tol, "dgamma")
        assert_close(norm.bias.grad.cpu(), bet.grad, tol, "dbeta")
    # a conv bias in front of a norm has a mathematically zero gradient: both sides are rounding noise
    assert conv.bias.grad.abs().max().item() < 1e-2 * (conv.weight.grad.abs().max().item() + 1e-6) + 1e-3


@pytest.mark.parametrize("act", ["tanh", "sigmoid", "none"])
def test_head_nchw_backward(act):
    """7x7 reflect heads writing planar NCHW with tanh / sigmoid / x20 scale (networks.py:178,182-183,212)."""
    from vid2vid_amd import lib as L
    from vid2vid_amd import autograd as AG
    torch.manual_seed(5)
    eng = _engine("fp32")
    cin, cout, H, W = 32, {"tanh": 3, "sigmoid": 1, "none": 2}[act], 16, 24
    scale = 20.0 if act == "none" else 1.0
    conv = nn.Conv2d(cin, cout, 7)
    with torch.no_grad():
        conv.weight.normal_(0, 0.05); conv.bias.normal_(0, 0.1)
    x = torch.randn(1, cin, H, W)
    xr = x.clone().requires_grad_(True)
    cref = nn.Conv2d(cin, cout, 7)
    cref.load_state_dict(conv.state_dict())
    raw = cref(F.pad(xr, (3,) * 4, mode="reflect"))
    yr = {"tanh": torch.tanh, "sigmoid": torch.sigmoid, "none": lambda t: t}[act](raw) * scale
    r = torch.randn_like(yr)
    (yr * r).sum().backward()
    conv = conv.to(DEV)
    xg = x.to(DEV).requires_grad_(True)
    code = {"tanh": L.ACT_TANH, "sigmoid": L.ACT_SIGMOID, "none": L.ACT_NONE}[act]
    y = AG.conv_group(eng, eng.pack(xg), conv, L.PAD_REFLECT, 3, None, code, 0.0, None, None, True, scale, "head")
    assert y.shape == yr.shape and y.dtype == torch.float32
    assert_close(y.detach().cpu(), yr.detach(), 1e-4, "forward")
    (y * r.to(DEV)).sum().backward()
    assert_close(xg.grad.cpu(), xr.grad, 2e-4, "dX")
    assert_close(conv.weight.grad.cpu(), cref.weight.grad, 2e-4, "dW")
    assert_close(conv.bias.grad.cpu(), cref.bias.grad, 2e-4, "db")


@pytest.mark.parametrize("shape", [(1, 5, 17, 23), (2, 8, 16, 32), (1, 3, 1, 9)])
def test_avgpool_nhwc_backward(shape):
    torch.manual_seed(3)
    eng = _engine("fp32")
    x = torch.randn(*shape)
    xr = x.clone().requires_grad_(True)
    yr = F.avg_pool2d(xr, 3, stride=2, padding=1, count_include_pad=False)
    r = torch.randn_like(yr)
    (yr * r).sum().backward()
    xg = x.to(DEV).requires_grad_(True)
    y = eng.unpack(eng.avgpool_nhwc(eng.pack(xg)))
    assert_close(y.detach().cpu(), yr.detach(), 1e-5, "forward")
    (y * r.to(DEV)).sum().backward()
    assert_close(xg.grad.cpu(), xr.grad, 1e-5, "dX")


def _ref_resample(img, flow):
    from oracle import vid2vid_oracle as O
    return O.resample(img, flow, align_corners=False)


def test_pack_concat_and_scale():
    from vid2vid_amd import autograd as AG
    torch.manual_seed(2)
    eng = _engine("fp32")
    a, b = torch.randn(2, 5, 7, 9), torch.randn(2, 2, 7, 9)
    ag = a.to(DEV).requires_grad_(True)
    x = AG.pack_concat(eng, ag, b.to(DEV), 1.0 / 20.0)
    y = eng.unpack(x)
    assert_close(y.detach().cpu(), torch.cat([a, b / 20.0], 1), 1e-6, "concat")
    r = torch.randn_like(y)
    (y * r).sum().backward()
    assert_close(ag.grad.cpu(), r[:, :5].cpu(), 1e-6, "d x0")


@pytest.mark.parametrize("with_fg", [False, True])
@pytest.mark.parametrize("prev_grad", [False, True])
def test_warp_blend_backward(with_fg, prev_grad):
    """Composite tail vs the reference expression (networks.py:216-230) under CPU autograd."""
    torch.manual_seed(17)
    eng = _engine("fp32")
    N, C, H, W = 1, 3, 20, 28
    raw = torch.tanh(torch.randn(N, C, H, W))
    flow = torch.randn(N, 2, H, W) * 3.0
    flow[0, 0, 0, :4] = 40.0            # clipped coordinates: no gradient to the flow there (ATen semantics)
    flow[0, 1, -1, :4] = -40.0
    wgt = torch.sigmoid(torch.randn(N, 1, H, W))
    prev = torch.tanh(F.interpolate(torch.randn(N, C, 5, 7), size=(H, W), mode="bilinear", align_corners=False))
    fg = torch.tanh(torch.randn(N, C, H, W))
    mask = (torch.rand(N, 1, H, W) > 0.6).float()
    ts = [raw, flow, wgt, prev, fg]
    tr = [t.clone().requires_grad_(True) for t in ts]
    warp = _ref_resample(tr[3], tr[1])
    fin = tr[0] * tr[2] + warp * (1 - tr[2])
    rawo = tr[0]
    if with_fg:
        fin = tr[4] * mask + fin * (1 - mask)
        rawo = tr[4] * mask + tr[0] * (1 - mask)
    r1, r2 = torch.randn_like(fin), torch.randn_like(fin)
    ((fin * r1).sum() + (rawo * r2).sum()).backward()

    tg = [t.to(DEV).requires_grad_(True) for t in ts]
    if not prev_grad:
        tg[3] = prev.to(DEV)
    (final, raw_blend), _ = eng.warp_blend(tg[0], tg[1], tg[2], tg[3], tg[4] if with_fg else None,
                                           mask.to(DEV) if with_fg else None)
    assert_close(final.detach().cpu(), fin.detach(), 1e-5, "final")
    assert_close(raw_blend.detach().cpu(), rawo.detach(), 1e-6, "raw blend")
    ((final * r1.to(DEV)).sum() + (raw_blend * r2.to(DEV)).sum()).backward()
    assert_close(tg[0].grad.cpu(), tr[0].grad, 1e-5, "d raw")
    assert_close(tg[1].grad.cpu(), tr[1].grad, 1e-4, "d flow")
    assert_close(tg[2].grad.cpu(), tr[2].grad, 1e-4, "d weight")
    if prev_grad:
        assert_close(tg[3].grad.cpu(), tr[3].grad, 1e-4, "d prev")
    if with_fg:
        assert_close(tg[4].grad.cpu(), tr[4].grad, 1e-5, "d fg")


def test_resample_backward():
    torch.manual_seed(23)
    eng = _engine("fp32")
    img = torch.randn(2, 3, 12, 16)
    flow = torch.randn(2, 2, 12, 16) * 2.5
    ir, fr = img.clone().requires_grad_(True), flow.clone().requires_grad_(True)
    out = _ref_resample(ir, fr)
    r = torch.randn_like(out)
    (out * r).sum().backward()
    ig, fg_ = img.to(DEV).requires_grad_(True), flow.to(DEV).requires_grad_(True)
    o = eng.resample_flow(ig, fg_)
    assert_close(o.detach().cpu(), out.detach(), 1e-5, "forward")
    (o * r.to(DEV)).sum().backward()
    assert_close(ig.grad.cpu(), ir.grad, 1e-4, "d img")
    assert_close(fg_.grad.cpu(), fr.grad, 1e-4, "d flow")


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_losses_forward_backward(prec):
    from vid2vid_amd import autograd as AG
    torch.manual_seed(29)
    eng = _engine(prec)
    rnd = (lambda t: t.bfloat16().float()) if prec == "bf16" else (lambda t: t.clone())
    tol = 1e-5 if prec == "fp32" else 1e-2
    # GANLoss / feature matching on NHWC activations (pad channels must not count)
    a, b = rnd(torch.randn(2, 5, 9, 11)), rnd(torch.randn(2, 5, 9, 11))
    ar = a.clone().requires_grad_(True)
    lr_ = F.mse_loss(ar, torch.ones_like(ar)) * 1.0 + F.l1_loss(ar, b) * 2.5
    lr_.backward()
    ag = a.to(DEV).requires_grad_(True)
    xa, xb = eng.pack(ag), eng.pack(b.to(DEV))
    l = AG.mse_const_act(eng, xa, 1.0) + AG.l1_act(eng, xa, xb, weight=2.5)
    assert l.shape == (1, 1)
    assert abs(l.item() - lr_.item()) <= tol * max(1.0, abs(lr_.item()))
    l.sum().backward()
    assert_close(ag.grad.cpu(), ar.grad, 1e-5 if prec == "fp32" else 2e-2, "d a (nhwc)")
    # MaskedL1Loss on planar tensors (always fp32 at the API)
    p, q = torch.randn(2, 3, 9, 11), torch.randn(2, 3, 9, 11)
    m = (torch.rand(2, 1, 9, 11) > 0.4).float()
    pr = p.clone().requires_grad_(True)
    me = m.expand(-1, 3, -1, -1)
    lm = F.l1_loss(pr * me, q * me) * 10.0
    lm.backward()
    pg = p.to(DEV).requires_grad_(True)
    l2 = AG.masked_l1(eng, pg, q.to(DEV), m.to(DEV), weight=10.0)
    assert abs(l2.item() - lm.item()) <= 1e-5 * max(1.0, abs(lm.item()))
    l2.sum().backward()
    assert_close(pg.grad.cpu(), pr.grad, 1e-5, "d a (masked planar)")


def test_fused_adam_matches_torch():
    from vid2vid_amd.optim import FusedAdam
    torch.manual_seed(31)
    shapes = [(7, 5, 3, 3), (13,), (4, 9)]
    ref_p = [nn.Parameter(torch.randn(*s)) for s in shapes]
    dev_p = [nn.Parameter(p.detach().clone().to(DEV)) for p in ref_p]
    ref_opt = torch.optim.Adam(ref_p, lr=2e-4, betas=(0.5, 0.999))
    opt = FusedAdam(dev_p, lr=2e-4, betas=(0.5, 0.999))
    for it in range(4):
        grads = [torch.randn(*s) for s in shapes]
        ref_opt.zero_grad(); opt.zero_grad()
        for p, g in zip(ref_p, grads):
            p.grad = g.clone()
        for p, g in zip(dev_p, grads):
            assert p.grad is not None and float(p.grad.abs().sum()) == 0.0      # zeroed flat views
            p.grad.add_(g.to(DEV))
        ref_opt.step(); opt.step()
    for p, q in zip(dev_p, ref_p):
        assert_close(p.detach().cpu(), q.detach(), 1e-6, "adam parameter")
    # TTUR variant: beta1 = 0 (vid2vid_model_G.py:78-80)
    p1, p2 = nn.Parameter(torch.ones(5)), nn.Parameter(torch.ones(5, device=DEV))
    o1, o2 = torch.optim.Adam([p1], lr=1e-3, betas=(0.0, 0.9)), FusedAdam([p2], lr=1e-3, betas=(0.0, 0.9))
    p1.grad = torch.arange(5.0); p2.grad.copy_(torch.arange(5.0).to(DEV))
    o1.step(); o2.step()
    assert_close(p2.detach().cpu(), p1.detach(), 1e-6, "adam ttur")


def test_fused_adam_capturable_replays_in_a_hipgraph():
    """optim.FusedAdam(capturable): step count / learning rate in device memory (v2v_adam_step_dev).  A step captured ONCE by
    stream capture and replayed N times equals N eager steps of torch.optim.Adam -- the bias corrections advance with the
    device counter, a learning-rate change on the host reaches the replays (sync_hyper), the checkpointed step count is the
    device's."""
    from vid2vid_amd.optim import FusedAdam
    torch.manual_seed(32)
    shapes = [(7, 5, 3, 3), (13,), (4, 9)]
    ref_p = [nn.Parameter(torch.randn(*s)) for s in shapes]
    dev_p = [nn.Parameter(p.detach().clone().to(DEV)) for p in ref_p]
    ref_opt = torch.optim.Adam(ref_p, lr=2e-4, betas=(0.5, 0.999))
    opt = FusedAdam(dev_p, lr=2e-4, betas=(0.5, 0.999))
    gbuf = [torch.zeros(*s, device=DEV) for s in shapes]                 # static gradient source of the captured step

    def body():
        opt.zero_grad()
        for p, g in zip(dev_p, gbuf):
            p.grad.add_(g)
        opt.step()

    def ref_step(grads):
        ref_opt.zero_grad()
        for p, g in zip(ref_p, grads):
            p.grad = g.clone()
        ref_opt.step()

    for it in range(2):                                                  # two eager steps first (host counter = 2)
        grads = [torch.randn(*s) for s in shapes]
        for b, g in zip(gbuf, grads):
            b.copy_(g)
        body(); ref_step(grads)
    opt.make_capturable()
    assert opt.device_step() == 2
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body()
    for it in range(5):
        if it == 3:                                                      # update_learning_rate (models/base_model.py:154-160)
            for o in (opt, ref_opt):
                o.param_groups[0]["lr"] = 5e-5
            opt.sync_hyper()
        grads = [torch.randn(*s) for s in shapes]
        for b, gr in zip(gbuf, grads):
            b.copy_(gr)
        g.replay(); ref_step(grads)
    torch.cuda.synchronize()
    assert opt.device_step() == 7 and opt.state_dict()["step"] == 7
    for p, q in zip(dev_p, ref_p):
        assert_close(p.detach().cpu(), q.detach(), 2e-6, "capturable adam parameter after 2 eager + 5 replayed steps")


def test_fused_adam_grad_scale_is_the_mean_of_summed_gradients():
    """v2v_adam_step(grad_scale = 1/world): the all-reduced SUM of the ranks' gradients is turned into the mean inside the
    optimizer kernel (parallel.GradSync) -- equal to torch.optim.Adam stepped on the averaged gradients."""
    import ctypes as C
    from vid2vid_amd.lib import lib, check
    from vid2vid_amd.optim import FusedAdam
    torch.manual_seed(77)
    world = 4
    ref_p = nn.Parameter(torch.randn(1000))
    dev_p = nn.Parameter(ref_p.detach().clone().to(DEV))
    ref_opt = torch.optim.Adam([ref_p], lr=2e-4, betas=(0.5, 0.999))
    opt = FusedAdam([dev_p], lr=2e-4, betas=(0.5, 0.999))

    class FakeSync:                       # what GradSync.all_reduce returns after summing `world` ranks
        world, force_collective = 1, False
        def wait_pending(self, device=None, owner=None): pass
        def all_reduce(self, flat): return 1.0 / world

    opt.grad_sync = FakeSync()
    for it in range(3):
        per_rank = [torch.randn(1000) for _ in range(world)]
        ref_opt.zero_grad(); opt.zero_grad()
        ref_p.grad = sum(per_rank) / world
        dev_p.grad.add_(sum(per_rank).to(DEV))            # the buffer holds the SUM after the all-reduce
        ref_opt.step(); opt.step()
    assert_close(dev_p.detach().cpu(), ref_p.detach(), 1e-6, "adam with grad_scale = 1/world")


def test_single_rank_rccl_overlapped_optimizer_step():
    """The multi-GPU path on the one GPU there is: a world-size-1 `nccl` (= RCCL) process group, parallel.sync_optimizers
    with the collective forced on.  FusedAdam.step enqueues the bucketed RCCL all-reduce and the Adam kernel on the side
    stream behind the compute stream's backward work and returns; parameters read after parallel.wait_pending() equal a
    plain (un-synchronised) FusedAdam's; zero_grad of the next step waits for the overlapped step by itself."""
    import os, socket
    import torch.distributed as dist
    from vid2vid_amd import parallel
    from vid2vid_amd.optim import FusedAdam
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        torch.manual_seed(3)
        shapes = [(64, 32, 3, 3), (64,), (300, 70)]
        a = [nn.Parameter(torch.randn(*sh, device=DEV)) for sh in shapes]
        b = [nn.Parameter(p.detach().clone()) for p in a]
        oa, ob = FusedAdam(a, lr=1e-3, betas=(0.5, 0.999)), FusedAdam(b, lr=1e-3, betas=(0.5, 0.999))
        gs = parallel.sync_optimizers([oa], bucket_bytes=16 << 10, force_collective=True)     # several buckets
        assert len(gs.buckets(oa.flat.flat_grad)) > 2 and dist.get_backend() == "nccl"
        for it in range(3):
            oa.zero_grad(); ob.zero_grad()
            for p, q in zip(a, b):
                g = torch.randn_like(p)
                p.grad.add_(g); q.grad.add_(g)
            oa.step(); ob.step()
            assert gs.pending(oa)                        # enqueued, not waited for
        parallel.wait_pending()
        torch.cuda.synchronize()
        for p, q in zip(a, b):
            assert torch.equal(p.detach(), q.detach()), "world 1: the all-reduce is the identity"
    finally:
        parallel._ACTIVE_SYNCS.clear()
        dist.destroy_process_group()


def test_two_optimizers_wait_only_for_their_own_overlapped_step():
    """train.py:86-93 order: optimizer_G.step() then optimizer_D.zero_grad().  Both optimizers share ONE GradSync; D's
    zero_grad must leave G's (all-reduce + Adam) pair pending on the side stream (it used to wait for all of
    them, which serialised G's 1.66 GB all-reduce in front of D's backward), G's own zero_grad must wait for it, and
    parallel.wait_pending() clears whatever is left."""
    import os, socket
    import torch.distributed as dist
    from vid2vid_amd import parallel
    from vid2vid_amd.optim import FusedAdam
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        torch.manual_seed(4)
        pg = [nn.Parameter(torch.randn(256, 64, 3, 3, device=DEV))]
        pd = [nn.Parameter(torch.randn(64, 16, 4, 4, device=DEV))]
        rg, rd = [nn.Parameter(p.detach().clone()) for p in pg], [nn.Parameter(p.detach().clone()) for p in pd]
        og, od = FusedAdam(pg, lr=1e-3), FusedAdam(pd, lr=1e-3)
        ref_g, ref_d = FusedAdam(rg, lr=1e-3), FusedAdam(rd, lr=1e-3)
        gs = parallel.sync_optimizers([og, od], bucket_bytes=64 << 10, force_collective=True)
        assert og.grad_sync is od.grad_sync is gs
        for it in range(3):
            og.zero_grad(); ref_g.zero_grad()
            assert not gs.pending(og)
            g = torch.randn_like(pg[0]); pg[0].grad.add_(g); rg[0].grad.add_(g)
            og.step(); ref_g.step()
            assert gs.pending(og)
            od.zero_grad(); ref_d.zero_grad()                 # D's zero_grad: G's pair stays in flight
            assert gs.pending(og) and not gs.pending(od)
            g = torch.randn_like(pd[0]); pd[0].grad.add_(g); rd[0].grad.add_(g)
            od.step(); ref_d.step()
            assert gs.pending(og) and gs.pending(od)
        og.zero_grad()                                        # its own event only
        assert not gs.pending(og) and gs.pending(od)
        parallel.wait_pending()
        assert not gs.pending()
        torch.cuda.synchronize()
        assert torch.equal(pg[0].detach(), rg[0].detach()) and torch.equal(pd[0].detach(), rd[0].detach())
    finally:
        parallel._ACTIVE_SYNCS.clear()
        dist.destroy_process_group()


def test_discard_stale_grads_follows_train_py_protocol():
    """optim.FusedAdam(discard_stale_grads): a "generator" conv and a "discriminator" conv stepped the way train.py:130-138 does it
    (zero_grad G / loss_G.backward through BOTH / step G, then zero_grad D / loss_D.backward on the detached fake / step D).  With the
    flag the discriminator's weight gradient of loss_G.backward() is not computed (no wgrad launch logged for it after D's first
    step) -- and the weights of both layers after three iterations are bit-identical to the run without the flag, where that
    gradient is computed and wiped by D's zero_grad()."""
    from vid2vid_amd import lib as L
    from vid2vid_amd import autograd as AG
    from vid2vid_amd.optim import FusedAdam

    def run(flag):
        torch.manual_seed(21)
        eng = _engine("fp32")
        g = nn.Conv2d(8, 8, 3, padding=1).to(DEV)
        d = nn.Conv2d(8, 4, 3, padding=1).to(DEV)
        og, od = FusedAdam(list(g.parameters()), lr=1e-2), FusedAdam(list(d.parameters()), lr=1e-2)
        og.discard_stale_grads = od.discard_stale_grads = flag
        x = torch.randn(1, 8, 12, 16, device=DEV)
        real = torch.randn(1, 8, 12, 16, device=DEV)
        d_wgrads_in_g_pass = []
        for it in range(3):
            fake = AG.conv_group(eng, eng.pack(x), g, L.PAD_ZERO, 1, None, L.ACT_RELU, 0.0, None, None, False, 1.0, "g")
            pred_fake = AG.conv_group(eng, fake, d, L.PAD_ZERO, 1, None, L.ACT_NONE, 0.0, None, None, False, 1.0, "d")
            loss_G = (eng.unpack(pred_fake) ** 2).mean()
            pf = AG.conv_group(eng, fake.detach(), d, L.PAD_ZERO, 1, None, L.ACT_NONE, 0.0, None, None, False, 1.0, "d")
            pr = AG.conv_group(eng, eng.pack(real), d, L.PAD_ZERO, 1, None, L.ACT_NONE, 0.0, None, None, False, 1.0, "d")
            loss_D = (eng.unpack(pf) ** 2).mean() + ((eng.unpack(pr) - 1) ** 2).mean()
            og.zero_grad()
            n0 = len(eng.conv_log)
            loss_G.backward()
            d_wgrads_in_g_pass.append(sum(1 for c in eng.conv_log[n0:] if c.get("kind") == "wgrad" and c["label"].endswith(":d")))
            og.step()
            od.zero_grad(); loss_D.backward(); od.step()
        torch.cuda.synchronize()
        return g.weight.detach().clone(), d.weight.detach().clone(), d.bias.detach().clone(), d_wgrads_in_g_pass

    gw0, dw0, db0, n_off = run(False)
    gw1, dw1, db1, n_on = run(True)
    assert n_off == [1, 1, 1], n_off
    assert n_on == [1, 0, 0], n_on                            # before D's first step its .grad is still what torch would hold
    assert torch.equal(gw0, gw1) and torch.equal(dw0, dw1) and torch.equal(db0, db1)


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("shape", [(1, 5, 16, 24), (2, 16, 17, 23), (1, 64, 8, 6), (1, 3, 2, 2)])
def test_maxpool2_forward_backward(shape, prec):
    """v2v_maxpool2_nhwc(+_backward) against F.max_pool2d autograd, incl. odd sizes (floor) and ties behind a ReLU
    (ATen keeps the FIRST maximal element of a window; the gradient must go to the same one)."""
    torch.manual_seed(5)
    eng = _engine(prec)
    x = torch.relu(torch.randn(*shape))                    # ~50 % exact zeros -> many all-zero windows (ties)
    x = (x * 8).round() / 8                                 # exactly representable in bf16, more ties
    xr = x.clone().requires_grad_(True)
    yr = F.max_pool2d(xr, 2, 2)
    r = ((torch.randn_like(yr) * 4).round() / 4)
    (yr * r).sum().backward()
    xg = x.to(DEV).requires_grad_(True)
    y = eng.unpack(eng.maxpool2_nhwc(eng.pack(xg)))
    assert torch.equal(y.detach().cpu(), yr.detach()), "forward"
    (y * r.to(DEV)).sum().backward()
    assert torch.equal(xg.grad.cpu(), xr.grad), "dX (argmax routing)"
    with torch.no_grad():                                   # inference path (no autograd Function)
        y2 = eng.unpack(eng.maxpool2_nhwc(eng.pack(x.to(DEV))))
    assert torch.equal(y2.cpu(), yr.detach())


@pytest.mark.parametrize("shape", [(2, 3, 16, 24), (1, 3, 9, 13)])
def test_avgpool2_planar_forward_backward(shape):
    torch.manual_seed(6)
    eng = _engine("fp32")
    x = torch.randn(*shape)
    xr = x.clone().requires_grad_(True)
    yr = F.avg_pool2d(xr, 2, stride=2, count_include_pad=False)
    r = torch.randn_like(yr)
    (yr * r).sum().backward()
    xg = x.to(DEV).requires_grad_(True)
    y = eng.avgpool2_planar(xg)
    assert_close(y.detach().cpu(), yr.detach(), 1e-6, "forward")
    (y * r.to(DEV)).sum().backward()
    assert_close(xg.grad.cpu(), xr.grad, 1e-6, "dX")
    with torch.no_grad():
        assert_close(eng.avgpool2_planar(x.to(DEV)).cpu(), yr.detach(), 1e-6, "forward (no grad)")


def test_onehot_planar_matches_encode_input():
    from oracle import vid2vid_oracle as O
    torch.manual_seed(7)
    eng = _engine("fp32")
    H, W = 19, 37
    lab = torch.randint(0, 35, (1, 1, 1, H, W)).float()
    inst = torch.randint(0, 4, (1, 1, 1, H // 4 + 1, W // 4 + 1)).float().repeat_interleave(4, 3).repeat_interleave(4, 4)[..., :H, :W]
    ref = O.encode_input(lab, inst.contiguous(), 35)[0, 0]
    got = eng.onehot_planar(lab[0, 0, 0].to(DEV).contiguous(), inst[0, 0, 0].to(DEV).contiguous(), H, W, 35)
    assert torch.equal(got.cpu(), ref)
    got = eng.onehot_planar(lab[0, 0, 0].to(DEV).contiguous(), None, H, W, 35)
    assert torch.equal(got.cpu(), ref[:35])
