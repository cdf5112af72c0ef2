"""PyTorch-ROCm custom ops (torch.autograd.Function) over libv2v_hip.so: the training path.

The reference trains through ATen autograd (cuDNN convolution backward, BatchNorm backward,
grid_sampler backward, L1/MSE loss backward).  Here every differentiable step of the hot path is
one Function whose forward AND backward are launches of the C-ABI library -- torch only keeps the
graph.  Conventions:

  * activations flow between ops as NHWC tensors [N,H,W,Cs] in the engine dtype (engine.Act);
    API-facing tensors (images, flows, weights, losses) are planar fp32 NCHW like the reference's;
  * parameter gradients are ACCUMULATED IN PLACE into `param.grad` by the kernels themselves
    (v2v_conv_wgrad / v2v_bn_backward / v2v_channel_sum with accumulate=1), so the Functions return
    None for parameters: no framework add kernel runs for the ~400 parameter tensors, and with
    optim.FusedAdam `.grad` is a view into one flat buffer that RCCL all-reduces in place;
  * nothing here falls back to torch compute.
"""
import ctypes as C
import os

import torch
import torch.nn as nn

from . import lib as L
from .lib import lib, check, WgradDesc, ConvDesc
from .engine import Act, pad_channels, _ptr, _stream, _TORCH_DTYPE
from .parallel import note_use, note_done


def _live(p):
    """False when a gradient written to this parameter now would be wiped unread: its FusedAdam (discard_stale_grads) has stepped
    and not been re-armed by zero_grad() -- the discriminators' parameters during loss_G.backward() (train.py:130-138)."""
    flat = getattr(p, "_v2v_flat", None)
    return True if flat is None else flat.live


def _grad_ptr(p):
    """Pointer of the parameter's gradient buffer (allocated zeroed on first use)."""
    if p is None or not p.requires_grad:
        return None
    if p.grad is None:
        p.grad = torch.zeros_like(p, memory_format=torch.contiguous_format)
    g = p.grad
    if g.dtype != torch.float32 or not (g.is_contiguous() or is_channels_last(g)):
        raise RuntimeError("parameter gradients must be fp32, contiguous or channels-last (optim.FlatBuffers)")
    return C.c_void_p(g.data_ptr())


_NORM_BIAS_GRAD = os.environ.get("V2V_NORM_BIAS_GRAD", "0") == "1"
# measurement only (scripts/gpu_r6.sh trainab): leave the weight-gradient launches out, to see what the rest of the step costs
# with nothing beside it on the chip.  The step it times is NOT a training step.
_SKIP_WGRAD = os.environ.get("V2V_SKIP_WGRAD", "0") == "1"


def is_channels_last(t):
    """A 4-D tensor stored [d0][KH][KW][d1] (optim.FlatBuffers keeps conv weights and their gradients that way)."""
    return t.dim() == 4 and not t.is_contiguous() and t.permute(0, 2, 3, 1).is_contiguous()


class _Cfg:
    """Static description of one conv group (not a tensor: passed through Function.apply untouched)."""
    __slots__ = ("eng", "conv", "norm", "pad_mode", "pad", "act", "act_param", "out_scale", "nchw", "label", "cin")


def conv_group(eng, x, conv, pad_mode, pad_override, norm, act, act_param, add0, add1, head_nchw, out_scale, label):
    cfg = _Cfg()
    cfg.eng, cfg.conv, cfg.norm = eng, conv, norm
    cfg.pad_mode = pad_mode
    cfg.pad = conv.padding[0] if pad_override is None else pad_override
    cfg.act, cfg.act_param = act, act_param
    cfg.out_scale = out_scale if head_nchw else 1.0
    cfg.nchw, cfg.label, cfg.cin = bool(head_nchw), label, x.C
    gamma = beta = None
    if norm is not None and getattr(norm, "affine", False):
        gamma, beta = norm.weight, norm.bias
    y = ConvFn.apply(cfg, x.t, conv.weight, conv.bias, gamma, beta,
                     None if add0 is None else add0.t, None if add1 is None else add1.t)
    return y if head_nchw else Act(y, conv.out_channels)


class ConvFn(torch.autograd.Function):
    """[ReflectionPad2d] Conv2d|ConvTranspose2d [BatchNorm2d|InstanceNorm2d, training statistics]
    [ReLU|LeakyReLU|Tanh|Sigmoid] [+ residual adds]  (models/networks.py:128-201, 554-593, 685-706)."""

    @staticmethod
    def forward(ctx, cfg, x_t, weight, bias, gamma, beta, add0_t, add1_t):
        eng = cfg.eng
        x = Act(x_t, cfg.cin)
        conv, norm = cfg.conv, cfg.norm
        cout = conv.out_channels
        ctx.cfg = cfg
        ctx.has_add = (add0_t is not None, add1_t is not None)
        # parameters whose .grad this node's backward accumulates into: counted per gradient bucket so that the data-parallel
        # runtime can all-reduce a bucket as soon as its last contribution of the pass is enqueued (parallel.BucketReady)
        ctx.grad_params = [t for t in (weight, bias, gamma, beta) if t is not None and t.requires_grad] if any(ctx.needs_input_grad) else []
        if ctx.grad_params:
            note_use(ctx.grad_params)
        if norm is not None:
            N, H, W = x.N, x.H, x.W
            pc = eng.packed(conv, x.Cs, refresh=False, touch=False)      # geometry only (a fetch with refresh re-packed the tap-major copy of every patch-tile layer once per step, on this stream)
            if pc.transposed:
                OH = (H - 1) * pc.stride - 2 * cfg.pad + pc.KH + pc.out_pad
                OW = (W - 1) * pc.stride - 2 * cfg.pad + pc.KW + pc.out_pad
            else:
                OH = (H + 2 * cfg.pad - pc.KH) // pc.stride + 1
                OW = (W + 2 * cfg.pad - pc.KW) // pc.stride + 1
            cs_raw = (cout + 3) // 4 * 4
            raw = torch.empty(N * OH * OW * cs_raw, dtype=torch.float32, device=eng.device)
            ss = torch.empty(4 * cout, dtype=torch.float32, device=eng.device)
            raw, rows, shp = eng.conv(x, conv, cfg.pad_mode, cfg.pad, L.OUT_RAW_F32_NHWC, want_stats=True, out=raw,
                                      label=cfg.label, fin=(norm, ss) if eng.fused_finalize else None)
            a0 = None if add0_t is None else Act(add0_t, cout)
            a1 = None if add1_t is None else Act(add1_t, cout)
            y = eng.norm_apply(raw, rows, shp, cout, norm, cfg.act, cfg.act_param, add0=a0, add1=a1,
                               label=cfg.label, ss=ss, finalized=eng.fused_finalize and eng.last_finalized)
            ctx.shape = shp
            ctx.save_for_backward(x_t, raw, ss)
            return y.t
        if add0_t is not None or add1_t is not None:
            raise NotImplementedError("residual adds need a norm layer in the group")
        out, _, shp = eng.conv(x, conv, cfg.pad_mode, cfg.pad, L.OUT_F32_NCHW if cfg.nchw else L.OUT_ACT_NHWC,
                               cfg.act, cfg.act_param, cfg.out_scale, label=cfg.label)
        y_t = out if cfg.nchw else out.t
        ctx.shape = shp
        ctx.save_for_backward(x_t, y_t if cfg.act != L.ACT_NONE else None, None)
        return y_t

    @staticmethod
    def backward(ctx, dy):
        cfg = ctx.cfg
        eng, conv, norm = cfg.eng, cfg.conv, cfg.norm
        x_t, s1, s2 = ctx.saved_tensors
        x = Act(x_t, cfg.cin)
        N, OH, OW = ctx.shape
        cout = conv.out_channels
        dt = eng.dtype
        cs_g = pad_channels(cout, dt)
        P = N * OH * OW
        st = _stream()
        dy = dy.contiguous()
        weight, bias = conv.weight, conv.bias
        # ---- gradient at the conv output (NHWC, engine dtype) ----
        if norm is not None:
            raw, ss = s1, s2
            g = torch.empty((N, OH, OW, cs_g), dtype=eng.tdtype, device=eng.device)
            rows = lib.v2v_bn_backward_rows(P)
            ws = eng.scratch("bn_bwd_ws", rows * 2 * cout + 2 * cout)
            affine = getattr(norm, "affine", False) and _live(norm.weight)
            check(lib.v2v_bn_backward(_ptr(dy), _ptr(raw), (cout + 3) // 4 * 4, _ptr(ss), _ptr(g), cs_g,
                                      _grad_ptr(norm.weight) if affine else None,
                                      _grad_ptr(norm.bias) if affine else None, 1, _ptr(ws),
                                      P, cout, dy.stride(2), cfg.act, cfg.act_param, dt, st), "bn_backward " + cfg.label)
        else:
            y_t = s1
            if cfg.act == L.ACT_NONE and cfg.out_scale == 1.0 and not cfg.nchw:
                g = dy
            else:
                g = torch.empty((N, OH, OW, cs_g), dtype=eng.tdtype, device=eng.device)
                check(lib.v2v_act_backward(_ptr(dy), _ptr(y_t), _ptr(g), N, OH, OW, cout,
                                           0 if cfg.nchw else dy.stride(2), cs_g, int(cfg.nchw), cfg.act,
                                           cfg.act_param, cfg.out_scale, dt, st), "act_backward " + cfg.label)
        cs_gs = g.stride(2)
        # ---- bias ----
        # A bias in front of a training-mode norm cancels in the norm's mean: its gradient, sum_p dRaw[p][c], is zero in exact
        # arithmetic (dRaw is mean-free by construction) and rounding noise of ~1e-7 x |dRaw| in the reference's autograd.  The
        # exact value is accumulated here (nothing is launched); V2V_NORM_BIAS_GRAD=1 restores the summed noise (one more
        # reduction launch per layer: 197 of the 434 channel-sum / norm-backward reductions of a 512x256 training chunk).
        live = _live(weight)                                  # (a layer's bias lives in the same flat buffer as its weight)
        if bias is not None and bias.requires_grad and norm is not None and not _NORM_BIAS_GRAD:
            _grad_ptr(bias)                                   # += 0: only makes sure the (zeroed) buffer exists
        elif bias is not None and bias.requires_grad and live:
            ws = eng.scratch("chsum_ws", lib.v2v_bn_backward_rows(P) * 2 * cout)
            check(lib.v2v_channel_sum(_ptr(g), _grad_ptr(bias), 1, _ptr(ws), P, cout, cs_gs, dt, st),
                  "channel_sum " + cfg.label)
        # ---- weight ----
        transposed = isinstance(conv, nn.ConvTranspose2d)
        if weight.requires_grad and not _SKIP_WGRAD and live:
            d = WgradDesc()
            if transposed:
                d.p, d.q = x_t.data_ptr(), g.data_ptr()
                d.N, d.OH, d.OW, d.QH, d.QW = N, x.H, x.W, OH, OW
                d.rows, d.cols, d.p_stride, d.q_stride = cfg.cin, cout, x.Cs, cs_gs
                d.stride, d.pad, d.pad_mode = 2, conv.padding[0], L.PAD_ZERO
            else:
                d.p, d.q = g.data_ptr(), x_t.data_ptr()
                d.N, d.OH, d.OW, d.QH, d.QW = N, OH, OW, x.H, x.W
                d.rows, d.cols, d.p_stride, d.q_stride = cout, cfg.cin, cs_gs, x.Cs
                d.stride, d.pad, d.pad_mode = conv.stride[0], cfg.pad, cfg.pad_mode
            d.KH, d.KW = conv.kernel_size
            d.grad = _grad_ptr(weight).value
            d.dtype, d.accumulate = dt, 1 + (2 if is_channels_last(weight.grad) else 0)
            d.zero_page = eng.zero_page().data_ptr()
            nbytes = lib.v2v_conv_wgrad_workspace(C.byref(d))
            if nbytes <= 0:
                check(int(nbytes) or -1, "conv_wgrad_workspace " + cfg.label)
            side = eng.wgrad_side_stream()
            if side is None:
                d.workspace = eng.scratch("wgrad_ws", (nbytes + 3) // 4).data_ptr()
                check(lib.v2v_conv_wgrad(C.byref(d), st), "conv_wgrad " + cfg.label)
            else:
                # dW is a leaf: its kernels run on the side stream, beside the serial norm-backward / backward-data chain
                # (engine.wgrad_side_stream).  Ordered behind everything on this stream so far (g, x, the zeroed .grad); the
                # operands are marked as in use there; the pass's end joins the streams (engine.queue_wgrad_join).
                side.wait_stream(torch.cuda.current_stream(eng.device))
                with torch.cuda.stream(side):
                    d.workspace = eng.scratch("wgrad_ws", (nbytes + 3) // 4).data_ptr()      # (slab scratch: used on this stream only)
                    check(lib.v2v_conv_wgrad(C.byref(d), C.c_void_p(side.cuda_stream)), "conv_wgrad " + cfg.label)
                g.record_stream(side)
                x_t.record_stream(side)
                eng.queue_wgrad_join()
            eng.log_backward("wgrad", cfg.label, conv, cfg.cin, cout, N, x.H * x.W if transposed else OH * OW)
        # ---- input ----
        dx = None
        if ctx.needs_input_grad[1]:
            dx = _conv_backward_data(eng, cfg, conv, transposed, g, cout, x, N, OH, OW)
        d0 = dy if ctx.has_add[0] and ctx.needs_input_grad[6] else None
        d1 = dy if ctx.has_add[1] and ctx.needs_input_grad[7] else None
        if ctx.grad_params:
            note_done(ctx.grad_params)                       # every kernel that writes these gradients is enqueued
        return None, dx, None, None, None, None, d0, d1


def _conv_backward_data(eng, cfg, conv, transposed, g, cout, x, N, OH, OW):
    """dX through the forward kernel with role-swapped weights (include/v2v_hip.h, v2v_conv2d)."""
    reflect = cfg.pad_mode == L.PAD_REFLECT
    # the generic operator's geometry (engine.PackedConv role 'bwd'); tune_backward_data fetches the packing the selected tile reads --
    # only that one is kept fresh after optimizer steps (Engine.repack_async walks the packings in use)
    bwd_pad = 0 if reflect else conv.padding[0]
    H, W = x.H, x.W
    p = cfg.pad
    HO, WO = (H + 2 * p, W + 2 * p) if reflect else (H, W)
    needs_zero = x.Cs != cfg.cin
    out = (torch.zeros if needs_zero else torch.empty)((N, HO, WO, x.Cs), dtype=eng.tdtype, device=eng.device)
    d = ConvDesc()
    d.in_, d.w, d.bias, d.out, d.stats = g.data_ptr(), None, None, out.data_ptr(), None
    d.zero_page = eng.zero_page().data_ptr()
    d.N, d.H, d.W = N, OH, OW
    d.cin, d.cin_stride, d.cout, d.cout_stride = cout, g.stride(2), cfg.cin, x.Cs
    d.KH, d.KW = conv.kernel_size
    d.stride = conv.stride[0]
    d.pad = bwd_pad
    d.pad_mode = L.PAD_ZERO
    d.transposed = int(not transposed)
    d.OH, d.OW = HO, WO
    d.dtype, d.out_mode, d.act, d.act_param, d.out_scale, d.tile = eng.dtype, L.OUT_ACT_NHWC, L.ACT_NONE, 0.0, 1.0, 0
    eng.tune_backward_data(d, cfg.cin, conv=conv, reflect=reflect)     # (also points d.w at the packing the selected tile reads)
    check(lib.v2v_conv2d(C.byref(d), _stream()), "conv backward-data " + cfg.label)
    eng.log_backward("bwd_data", cfg.label, conv, cfg.cin, cout, N, H * W if transposed else OH * OW)
    if not reflect:
        return out
    dx = torch.empty((N, H, W, x.Cs), dtype=eng.tdtype, device=eng.device)
    check(lib.v2v_reflect_pad_fold(_ptr(out), _ptr(dx), N, H, W, p, x.Cs, eng.dtype, _stream()), "reflect_pad_fold")
    return dx


# --------------------------------------------------------------------------------------
# layout / elementwise ops
# --------------------------------------------------------------------------------------
class PackFn(torch.autograd.Function):
    """cat([x0, x1], 1) (x1 optional) of planar fp32 NCHW -> NHWC engine dtype."""

    @staticmethod
    def forward(ctx, eng, x0, x1, scale1=1.0):
        x0 = x0.contiguous().float()
        N, C0, H, W = x0.shape
        C1 = 0
        if x1 is not None:
            x1 = x1.contiguous().float()
            C1 = x1.shape[1]
        cs = pad_channels(C0 + C1, eng.dtype)
        y = torch.empty((N, H, W, cs), dtype=eng.tdtype, device=eng.device)
        check(lib.v2v_pack_concat_nhwc(_ptr(x0), C0, _ptr(x1), C1, float(scale1), _ptr(y), N, H, W, cs, eng.dtype,
                                       _stream()), "pack_concat")
        if x1 is not None and ctx.needs_input_grad[2] and scale1 != 1.0:
            raise NotImplementedError("pack_concat: a scaled second operand is never differentiated on this path")
        ctx.eng, ctx.dims = eng, (N, C0, C1, H, W, cs)
        return y

    @staticmethod
    def backward(ctx, dy):
        eng = ctx.eng
        N, C0, C1, H, W, cs = ctx.dims
        dy = dy.contiguous()
        d0 = d1 = None
        if ctx.needs_input_grad[1]:
            d0 = torch.empty((N, C0, H, W), dtype=torch.float32, device=eng.device)
            check(lib.v2v_unpack_channels_nchw(_ptr(dy), _ptr(d0), N, C0, H, W, cs, 0, eng.dtype, _stream()), "unpack_channels")
        if C1 and ctx.needs_input_grad[2]:
            d1 = torch.empty((N, C1, H, W), dtype=torch.float32, device=eng.device)
            check(lib.v2v_unpack_channels_nchw(_ptr(dy), _ptr(d1), N, C1, H, W, cs, C0, eng.dtype, _stream()), "unpack_channels")
        return None, d0, d1, None


def pack_concat(eng, x0, x1=None, scale1=1.0):
    """Act of cat([x0, x1 * scale1], dim=1); differentiable w.r.t. x0 (and x1 when scale1 == 1)."""
    c = x0.shape[1] + (0 if x1 is None else x1.shape[1])
    return Act(PackFn.apply(eng, x0, x1, scale1), c)


class UnpackFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, eng, y, Cc):
        N, H, W, cs = y.shape[0], y.shape[1], y.shape[2], y.stride(2)
        x = torch.empty((N, Cc, H, W), dtype=torch.float32, device=eng.device)
        check(lib.v2v_unpack_channels_nchw(_ptr(y), _ptr(x), N, Cc, H, W, cs, 0, eng.dtype, _stream()), "unpack")
        ctx.eng, ctx.dims = eng, (N, Cc, H, W, cs)
        return x

    @staticmethod
    def backward(ctx, dx):
        eng = ctx.eng
        N, Cc, H, W, cs = ctx.dims
        dx = dx.contiguous().float()
        dy = torch.empty((N, H, W, cs), dtype=eng.tdtype, device=eng.device)
        check(lib.v2v_pack_concat_nhwc(_ptr(dx), Cc, None, 0, 1.0, _ptr(dy), N, H, W, cs, eng.dtype, _stream()), "pack")
        return None, dy, None


class AddFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, eng, a, b):
        y = torch.empty_like(a)
        check(lib.v2v_add_nhwc(_ptr(a), _ptr(b), _ptr(y), a.numel(), eng.dtype, _stream()), "add")
        return y

    @staticmethod
    def backward(ctx, dy):
        return None, dy, dy


class AvgPoolFn(torch.autograd.Function):
    """AvgPool2d(3, 2, 1, count_include_pad=False) on NHWC (MultiscaleDiscriminator.downsample, networks.py:652)."""

    @staticmethod
    def forward(ctx, eng, x):
        N, H, W, cs = x.shape[0], x.shape[1], x.shape[2], x.stride(2)
        OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        y = torch.empty((N, OH, OW, cs), dtype=x.dtype, device=x.device)
        check(lib.v2v_avgpool3s2_nhwc(_ptr(x), _ptr(y), N, H, W, cs, eng.dtype, _stream()), "avgpool")
        ctx.eng, ctx.dims = eng, (N, H, W, cs)
        return y

    @staticmethod
    def backward(ctx, dy):
        eng = ctx.eng
        N, H, W, cs = ctx.dims
        dy = dy.contiguous()
        dx = torch.empty((N, H, W, cs), dtype=dy.dtype, device=dy.device)
        check(lib.v2v_avgpool3s2_nhwc_backward(_ptr(dy), _ptr(dx), N, H, W, cs, eng.dtype, _stream()), "avgpool_backward")
        return None, dx


class MaxPool2Fn(torch.autograd.Function):
    """MaxPool2d(2, 2) on NHWC (torchvision VGG19 `features` 4 / 9 / 18 / 27 inside Vgg19, networks.py:840-870)."""

    @staticmethod
    def forward(ctx, eng, x):
        N, H, W, cs = x.shape[0], x.shape[1], x.shape[2], x.stride(2)
        y = torch.empty((N, H // 2, W // 2, cs), dtype=x.dtype, device=x.device)
        check(lib.v2v_maxpool2_nhwc(_ptr(x), _ptr(y), N, H, W, cs, eng.dtype, _stream()), "maxpool2")
        ctx.eng, ctx.dims = eng, (N, H, W, cs)
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        eng = ctx.eng
        N, H, W, cs = ctx.dims
        (x,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty((N, H, W, cs), dtype=dy.dtype, device=dy.device)
        check(lib.v2v_maxpool2_nhwc_backward(_ptr(dy), _ptr(x), _ptr(dx), N, H, W, cs, eng.dtype, _stream()),
              "maxpool2_backward")
        return None, dx


class AvgPool2PlanarFn(torch.autograd.Function):
    """AvgPool2d(2, stride 2, count_include_pad=False) on planar fp32 (VGGLoss.downsample, networks.py:782-786)."""

    @staticmethod
    def forward(ctx, eng, x):
        x = x.contiguous().float()
        H, W = x.shape[-2], x.shape[-1]
        planes = x.numel() // (H * W)
        y = torch.empty(tuple(x.shape[:-2]) + (H // 2, W // 2), dtype=torch.float32, device=x.device)
        check(lib.v2v_avgpool2_planar(_ptr(x), _ptr(y), planes, H, W, _stream()), "avgpool2")
        ctx.dims = (tuple(x.shape), planes, H, W)
        return y

    @staticmethod
    def backward(ctx, dy):
        shape, planes, H, W = ctx.dims
        dy = dy.contiguous().float()
        dx = torch.empty(shape, dtype=torch.float32, device=dy.device)
        check(lib.v2v_avgpool2_planar_backward(_ptr(dy), _ptr(dx), planes, H, W, _stream()), "avgpool2_backward")
        return None, dx


class WarpBlendFn(torch.autograd.Function):
    """Composite tail (models/networks.py:216-230): returns (img_final, blended img_raw)."""

    @staticmethod
    def forward(ctx, eng, img_raw, flow, weight, prev, fg, mask):
        N, Cc, H, W = img_raw.shape
        gx, gy = eng.grid(H, W)
        tensors = [None if t is None else t.contiguous().float() for t in (img_raw, flow, weight, prev, fg, mask)]
        img_raw, flow, weight, prev, fg, mask = tensors
        raw_blend = torch.empty_like(img_raw)
        check(lib.v2v_memcpy_d2d(_ptr(raw_blend), _ptr(img_raw), img_raw.numel() * 4, _stream()), "memcpy")
        final = torch.empty_like(img_raw)
        check(lib.v2v_warp_blend(_ptr(raw_blend), _ptr(flow), _ptr(weight), _ptr(prev), _ptr(fg), _ptr(mask),
                                 _ptr(final), None, _ptr(gx), _ptr(gy), N, Cc, H, W, int(eng.align_corners), _stream()),
              "warp_blend")
        ctx.eng = eng
        ctx.save_for_backward(img_raw, flow, weight, prev, fg, mask)
        return final, raw_blend

    @staticmethod
    def backward(ctx, d_final, d_rawout):
        eng = ctx.eng
        img_raw, flow, weight, prev, fg, mask = ctx.saved_tensors
        N, Cc, H, W = img_raw.shape
        gx, gy = eng.grid(H, W)
        d_final = d_final.contiguous()
        d_rawout = None if d_rawout is None else d_rawout.contiguous()
        dev = img_raw.device
        d_raw = torch.empty_like(img_raw)
        d_flow = d_weight = d_prev = d_fg = None
        if flow is not None:
            d_flow, d_weight = torch.empty_like(flow), torch.empty_like(weight)
            if ctx.needs_input_grad[4]:
                d_prev = torch.zeros_like(prev)
        if fg is not None:
            d_fg = torch.empty_like(fg)
        check(lib.v2v_warp_blend_backward(_ptr(d_final), _ptr(d_rawout), _ptr(img_raw), _ptr(flow), _ptr(weight),
                                          _ptr(prev), _ptr(fg), _ptr(mask), _ptr(gx), _ptr(gy), _ptr(d_raw),
                                          _ptr(d_flow), _ptr(d_weight), _ptr(d_prev), _ptr(d_fg), N, Cc, H, W,
                                          int(eng.align_corners), _stream()), "warp_blend_backward")
        return None, d_raw, d_flow, d_weight, d_prev, d_fg, None


class ResampleFn(torch.autograd.Function):
    """BaseNetwork.resample / BaseModel.resample (networks.py:108-115, base_model.py:189-196)."""

    @staticmethod
    def forward(ctx, eng, img, flow):
        img, flow = img.contiguous().float(), flow.contiguous().float()
        N, Cc, H, W = img.shape
        gx, gy = eng.grid(H, W)
        out = torch.empty_like(img)
        check(lib.v2v_resample_flow(_ptr(img), _ptr(flow), _ptr(out), _ptr(gx), _ptr(gy), N, Cc, H, W,
                                    int(eng.align_corners), _stream()), "resample_flow")
        ctx.eng = eng
        ctx.save_for_backward(img, flow)
        return out

    @staticmethod
    def backward(ctx, d_out):
        eng = ctx.eng
        img, flow = ctx.saved_tensors
        N, Cc, H, W = img.shape
        gx, gy = eng.grid(H, W)
        d_out = d_out.contiguous()
        d_img = torch.zeros_like(img) if ctx.needs_input_grad[1] else None
        d_flow = torch.empty_like(flow) if ctx.needs_input_grad[2] else None
        if d_img is None and d_flow is None:
            return None, None, None
        check(lib.v2v_resample_flow_backward(_ptr(d_out), _ptr(img), _ptr(flow), _ptr(gx), _ptr(gy), _ptr(d_img),
                                             _ptr(d_flow), N, Cc, H, W, int(eng.align_corners), _stream()),
              "resample_flow_backward")
        return None, d_img, d_flow


# --------------------------------------------------------------------------------------
# losses
# --------------------------------------------------------------------------------------
class LossFn(torch.autograd.Function):
    """weight * mean-reduced loss as a (1,1) fp32 tensor; gradient to `a` only (targets are detached in the
    reference: vid2vid_model_D.py:140,211; networks.py:768-771)."""

    @staticmethod
    def forward(ctx, eng, kind, a, b, mask, target, weight, C_real, planar):
        st = _stream()
        out = torch.empty((1, 1), dtype=torch.float32, device=eng.device)
        ws = eng.scratch("loss_ws", lib.v2v_loss_workspace_floats())
        if planar:
            a = a.contiguous().float()
            b = None if b is None else b.contiguous().float()
            mask = None if mask is None else mask.contiguous().float()
            N = a.shape[0]
            HW = a.shape[-2] * a.shape[-1]
            CHW = a.numel() // N
            dims = (0, 0, 0, N, CHW, HW)
        else:
            a = a.contiguous()
            b = None if b is None else b.contiguous()
            if b is not None and b.shape != a.shape:
                raise RuntimeError("loss: shape mismatch")
            P = a.shape[0] * a.shape[1] * a.shape[2]
            dims = (P, C_real, a.stride(2), 0, 0, 0)
        check(lib.v2v_loss_forward(kind, _ptr(a), _ptr(b), _ptr(mask), float(target), float(weight), *dims, int(planar),
                                   _ptr(ws), _ptr(out), eng.dtype, st), "loss_forward")
        ctx.eng, ctx.args = eng, (kind, float(target), float(weight), dims, int(planar))
        ctx.save_for_backward(a, b, mask)
        return out

    @staticmethod
    def backward(ctx, gout):
        eng = ctx.eng
        kind, target, weight, dims, planar = ctx.args
        a, b, mask = ctx.saved_tensors
        gout = gout.contiguous().float()
        da = torch.empty_like(a)
        check(lib.v2v_loss_backward(kind, _ptr(a), _ptr(b), _ptr(mask), target, weight, *dims, planar,
                                    _ptr(gout), _ptr(da), eng.dtype, _stream()), "loss_backward")
        return None, None, da, None, None, None, None, None, None


def mse_const_act(eng, pred, target, weight=1.0):
    """nn.MSELoss(pred, const) on an NHWC Act (GANLoss, networks.py:764-774)."""
    return LossFn.apply(eng, L.LOSS_MSE_CONST, pred.t, None, None, target, weight, pred.C, False)


def l1_act(eng, a, b, weight=1.0):
    """nn.L1Loss(a, b.detach()) between two NHWC Acts (criterionFeat)."""
    return LossFn.apply(eng, L.LOSS_L1, a.t, b.t.detach(), None, 0.0, weight, a.C, False)


def masked_l1(eng, a, b, mask, weight=1.0):
    """MaskedL1Loss (networks.py:804-812) on planar fp32 NCHW; gradient flows to `a` only."""
    return LossFn.apply(eng, L.LOSS_L1, a, b.detach(), None if mask is None else mask.detach(), 0.0, weight, 0, True)
