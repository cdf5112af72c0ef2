"""Lowering of the reference's nn.Module graphs onto libv2v_hip.so.

The reference describes every network as nn.Sequential lists of Conv2d / ConvTranspose2d /
ReflectionPad2d / BatchNorm2d / InstanceNorm2d / ReLU / LeakyReLU / Tanh / Sigmoid /
ResnetBlock (models/networks.py:117-725).  `Engine.run_sequential` walks such a list and
emits the fused HIP launches:

    [ReflectionPad2d(p)] Conv2d|ConvTranspose2d  ->  v2v_conv2d (padding folded into the loader)
       + norm [+ act]                            ->  conv(raw + statistics), v2v_bn_finalize,
                                                     v2v_bn_apply (normalise + act + residual adds)
       + act only / nothing                      ->  activation in the conv epilogue
    ResnetBlock                                   ->  two of the above, residual in bn_apply

All tensors handed to the library are NHWC with a channel stride padded to 16 bytes.  The
emitted launches either run immediately (eager) or are recorded into a `Plan` that is then
replayed per frame as one native call (hipGraphs, one per lane segment).  No torch compute op is on this path.
"""
import contextlib
import ctypes as C
import json
import math
import os
import sys

import torch
import torch.nn as nn

from . import lib as L
from .lib import lib, check, ConvDesc

_TORCH_DTYPE = {L.F32: torch.float32, L.BF16: torch.bfloat16}


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


_HAS_GPU = []


def _stream():
    """hipStream_t of torch's current stream.  (torch.cuda.is_available() re-reads the environment on every call: 4400 calls =
    ~8 ms of a 75 ms host-bound training chunk, scripts/host_profile_train.py -- its answer cannot change, so it is asked once.)"""
    if not _HAS_GPU:
        _HAS_GPU.append(torch.cuda.is_available())
    if not _HAS_GPU[0]:
        return None          # record-only dry runs on a CPU host never launch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def vec_of(dtype):
    return 8 if dtype == L.BF16 else 4


def pad_channels(c, dtype):
    v = vec_of(dtype)
    return (c + v - 1) // v * v


class Act:
    """NHWC activation: tensor [N,H,W,Cs] (Cs = padded channel stride), C real channels."""
    __slots__ = ("t", "C", "onehot", "x3")

    def __init__(self, t, C, onehot=None):
        self.t, self.C = t, C
        self.onehot = onehot      # LabelSource when this Act is the one-hot encoding of label maps (encode_labels)
        self.x3 = None            # x3 engines: the same values as a bf16x3 operand [hi | lo | hi], written by the producing bn_apply

    def detach(self):
        return Act(self.t.detach(), self.C)

    @property
    def N(self): return self.t.shape[0]
    @property
    def H(self): return self.t.shape[1]
    @property
    def W(self): return self.t.shape[2]
    @property
    def Cs(self): return self.t.stride(2)      # true channel stride (a channel-sliced view keeps its parent's)


class LabelSource:
    """The label / instance maps an encoded one-hot Act was built from: [T][H][W] fp32-encoded integers or uint8 / int32.
    A 7x7 stem convolution over such an Act runs as a weight gather-sum on the maps (csrc/onehot_stem.hip)."""
    __slots__ = ("labels", "inst", "T", "label_nc", "codes")

    def __init__(self, labels, inst, T, label_nc):
        self.labels, self.inst, self.T, self.label_nc = labels, inst, T, label_nc
        self.codes = None         # [T][H][W] uint8 label | edge << 7 (Engine.label_codes): what the stem kernels stage from


_PARAM_EPOCH = [0]


def bump_param_epoch():
    """Force every packed weight copy to be refreshed on next use (parameters changed behind torch's back by
    something other than optim.FusedAdam, which bumps the per-buffer counter of its own parameters instead -- frozen
    networks such as FlowNet2 / VGG19 are then never re-packed by a training step)."""
    _PARAM_EPOCH[0] += 1


_ENGINES = []        # weak references to every live Engine (repack_after_step walks them)


def repack_after_step(flat=None):
    """optim.FusedAdam.step, after its update kernel: re-pack -- on each engine's side stream -- every packed weight copy that
    was in use and is now stale.  Lazily, the training step re-packed each layer in front of its first use of the next chunk:
    ~270 small launches per 512x256 chunk in the serial chain of the forward / backward passes (5 % of the step's kernel time
    plus their launch turnarounds, profiles/r06_v1_train_kernel_stats.txt).  Issued here they run beside the next backward
    pass; the compute stream waits for them once, at the engine's next convolution (Engine.wait_repack)."""
    for ref in list(_ENGINES):
        eng = ref()
        if eng is None:
            _ENGINES.remove(ref)
        else:
            eng.repack_async(flat)


class PackedConv:
    """Device-resident packed weights of one nn.Conv2d / nn.ConvTranspose2d.

    role 'fwd':   the layer's own operator.
    role 'bwd':   its backward-data operator -- the SAME parameter tensor read with the roles of
                  cin / cout swapped (include/v2v_hip.h, v2v_conv2d): a Conv2d becomes a transposed conv
                  (stride s; pad p, or pad 0 behind a ReflectionPad2d), a ConvTranspose2d a stride-2 Conv2d.
    """

    def __init__(self, eng, mod, cin_stride, role="fwd", reflect=False, korder=0):
        self.mod = mod
        self.role = role
        self.korder = korder      # 0: tap-major K (implicit-GEMM tiles), 1: channel-chunk-major (patch kernel), 3: paired-x (PairedXConv)
        self.src_mod = None       # korder 3: the geometry below is the layer's own, the packed matrix its PairedXConv's (64 x 9 x 64, chunk-major)
        if korder == 3:
            if role != "fwd" or cin_stride != 32:
                raise ValueError("paired-x packing: forward operator of a layer with 64-byte pixels")
            self.src_mod = PairedXConvT(mod) if isinstance(mod, nn.ConvTranspose2d) else PairedXConv(mod)
        mod_t = getattr(mod, "is_transposed", isinstance(mod, nn.ConvTranspose2d))
        w = mod.weight
        self.KH, self.KW = mod.kernel_size
        self.stride = mod.stride[0]
        if mod_t:
            m_cin, m_cout = w.shape[0], w.shape[1]
            if tuple(mod.stride) != (2, 2):
                raise NotImplementedError("ConvTranspose2d must be stride 2")
        else:
            m_cout, m_cin = w.shape[0], w.shape[1]
        if role == "fwd":
            self.transposed = mod_t
            self.cin, self.cout = m_cin, m_cout
            self.pad = mod.padding[0]
        else:
            self.transposed = not mod_t
            self.cin, self.cout = m_cout, m_cin          # dY channels in, dX channels out
            self.pad = 0 if reflect else mod.padding[0]
        self.out_pad = mod.output_padding[0] if mod_t else 0
        self.cin_stride = cin_stride
        self.dtype = eng.dtype
        if korder == 4 and (role != "bwd" or mod_t or (self.KH, self.KW) != (3, 3) or self.stride != 1):
            raise ValueError("korder 4: the backward-data operator of a 3x3 / stride 1 Conv2d, packed as a convolution with flipped taps")
        if korder == 5 and (role != "bwd" or mod_t or self.KH != self.KW or self.KH % 2 == 0 or self.stride != 1):
            raise ValueError("korder 5: the backward-data operator of a square stride-1 Conv2d, packed tap-major as a convolution with flipped taps")
        n = ((lib.v2v_conv_packed_elems(64, 64, 32, 3, 3, 1, 2, 1, eng.dtype) if mod_t else lib.v2v_conv_packed_elems(64, 64, 64, 3, 3, 0, 1, self.pad, eng.dtype))
             if korder == 3 else
             lib.v2v_conv_packed_elems(self.cin, cin_stride, self.cout, 3, 3, 0, 1, 1, eng.dtype) if korder == 4 else
             lib.v2v_conv_packed_elems(self.cin, cin_stride, self.cout, self.KH, self.KW, 0, 1, self.KH // 2, eng.dtype) if korder == 5 else
             lib.v2v_conv_packed_elems(self.cin, cin_stride, self.cout, self.KH, self.KW,
                                       int(self.transposed), self.stride, self.pad, eng.dtype))
        self.buf = torch.empty(n, dtype=_TORCH_DTYPE[eng.dtype], device=eng.device)
        self.bias = None
        self.version = None
        self.used = True          # fetched since the last asynchronous re-pack (Engine.repack_async only refreshes what is in use)
        # the flat optimizer buffer that owns this layer's parameter (None: not managed by a FusedAdam, or a composed view such as
        # MergedConv whose `weight` is assembled on read): repack_after_step walks only the packings of the buffer that was stepped --
        # an engine shared by several models (bench.py builds a dozen) holds thousands of packings
        w_own = mod.__dict__.get("_parameters", {}).get("weight") if isinstance(mod, nn.Module) else None
        self.flat_id = id(getattr(w_own, "_v2v_flat", None)) if w_own is not None and getattr(w_own, "_v2v_flat", None) is not None else None
        self.refresh()

    def refresh(self, force=False):
        """(Re)pack if the parameter changed (optimizer step / load_state_dict)."""
        if self.src_mod is not None:               # paired-x: the 64 -> 64 matrix of the PairedXConv, as a chunk-major (korder 1) packing
            ver = self.src_mod.version_key()
            if not force and ver == self.version:
                return
            w32 = self.src_mod.weight.contiguous()
            if self.src_mod.is_transposed:           # 64 -> 32 transposed: the full-tap chunk-major matrix (korder 2) tile 114 reads
                check(lib.v2v_conv_pack_weights(_ptr(w32), _ptr(self.buf), 64, 64, 32, 3, 3, 1, 2, 1, self.dtype, 2, _stream()),
                      "conv_pack_weights (paired-x, transposed)")
            else:
                check(lib.v2v_conv_pack_weights(_ptr(w32), _ptr(self.buf), 64, 64, 64, 3, 3, 0, 1, self.pad, self.dtype, 1, _stream()),
                      "conv_pack_weights (paired-x)")
            self.bias = None if self.mod.bias is None else self.mod.bias.detach().float().contiguous()      # the layer's own 32 values (the kernel folds the index)
            self.version = ver
            return
        if hasattr(self.mod, "version_key"):       # MergedConv: `weight` is a fresh torch.cat on every read (version 0, recycled
            ver = self.mod.version_key()           # address) -- the key comes from the two source layers
            if not force and ver == self.version:
                return
            w = self.mod.weight
        else:
            w = self.mod.weight
            ver = _param_version(self.mod)
            if not force and ver == self.version:
                return
        w32 = w.detach()
        src_cl = (w32.dtype == torch.float32 and w32.dim() == 4 and not w32.is_contiguous()
                  and w32.permute(0, 2, 3, 1).is_contiguous())          # optim.FlatBuffers: channels-last master weights
        if self.korder in (4, 5):
            src_cl = False                                               # (the flipped packing reads the standard layout only)
        if not src_cl and (w32.dtype != torch.float32 or not w32.is_contiguous()):
            w32 = w32.float().contiguous()
        check(lib.v2v_conv_pack_weights(_ptr(w32), _ptr(self.buf), self.cin, self.cin_stride, self.cout,
                                        self.KH, self.KW, int(self.transposed), self.stride, self.pad, self.dtype,
                                        self.korder + (256 if src_cl else 0), _stream()),
              "conv_pack_weights")
        if self.role == "fwd":
            self.bias = None if self.mod.bias is None else self.mod.bias.detach().float().contiguous()
        self.version = ver


def _param_version(mod):
    """What changes when a layer's weight / bias changes: torch's version counters (in-place writes, load_state_dict),
    the storage address (re-homing into flat buffers), and the update counter of the fused optimizer that owns the
    parameter (optim.FlatBuffers.epoch: its HIP kernel writes are invisible to torch)."""
    w = mod.weight
    box = getattr(w, "_v2v_epoch", None)
    return (w._version, w.data_ptr(), None if mod.bias is None else mod.bias._version, _PARAM_EPOCH[0],
            0 if box is None else box[0])


class MergedConv:
    """Two Conv2d heads of identical geometry reading the same input (model_final_flow / model_final_w,
    models/networks.py:181-183) presented as ONE layer with the output channels concatenated: what PackedConv / Engine.conv
    need of an nn.Conv2d, with `weight` / `bias` assembled from the two source layers on every read (eager use re-packs;
    a recorded plan packs once, like every other layer)."""

    def __init__(self, a, b):
        if (a.kernel_size, a.stride, a.padding, a.in_channels, a.groups) != (b.kernel_size, b.stride, b.padding, b.in_channels, b.groups):
            raise ValueError("merged heads must have the same geometry")
        self.a, self.b = a, b
        self.kernel_size, self.stride, self.padding, self.groups = a.kernel_size, a.stride, a.padding, a.groups
        self.in_channels, self.out_channels = a.in_channels, a.out_channels + b.out_channels
        self.output_padding = (0, 0)

    def version_key(self):
        return _param_version(self.a) + _param_version(self.b)

    @property
    def weight(self):
        return torch.cat([self.a.weight.detach(), self.b.weight.detach()], 0)

    @property
    def bias(self):
        if self.a.bias is None and self.b.bias is None:
            return None
        z = lambda m: m.bias.detach() if m.bias is not None else torch.zeros(m.out_channels, device=m.weight.device)
        return torch.cat([z(self.a), z(self.b)], 0)


class X3Conv:
    """An nn.Conv2d seen by the bf16 kernels as a convolution over the bf16x3 operand (csrc/pointwise.hip, split_x3): the
    input channels are [hi(x) | lo(x) | hi(x)], the weights [hi(W) | hi(W) | lo(W)] -- the product is x * W to ~2^-17
    relative on the bf16 matrix pipe.  What PackedConv / Engine.conv need of an nn.Conv2d, assembled on read."""

    def __init__(self, conv):
        self.src = conv
        self.is_transposed = isinstance(conv, nn.ConvTranspose2d)      # weight [cin][cout][kh][kw]: the input channels are dim 0
        self.kernel_size, self.stride, self.padding, self.groups = conv.kernel_size, conv.stride, conv.padding, conv.groups
        self.in_channels, self.out_channels = 3 * conv.in_channels, conv.out_channels
        self.output_padding = tuple(conv.output_padding) if self.is_transposed else (0, 0)

    def version_key(self):
        return _param_version(self.src)

    @property
    def weight(self):
        w = self.src.weight.detach().float()
        hi = w.bfloat16().float()
        lo = (w - hi).bfloat16().float()
        return torch.cat([hi, hi, lo], 0 if self.is_transposed else 1).contiguous()

    @property
    def bias(self):
        return None if self.src.bias is None else self.src.bias.detach()


class PairedXConv:
    """A 3x3 / stride 1 nn.Conv2d with <= 32 input and exactly 32 output channels, seen as a 64 -> 64 convolution over PAIRS of
    horizontally adjacent pixels: the NHWC tensor [H][W][32] IS [H][W/2][64] (paired pixel X = pixels 2X, 2X+1), and
        out[y][X][a*32+co] = sum W'[a*32+co][b*32+ci][ky][kX] in[y+ky-1][X+kX-1][b*32+ci],   W'[..][ky][kX] = W[co][ci][ky][2kX+b-a-1]
    (zero where that tap index falls outside 0..2).  That puts the 32-channel ResnetBlocks of the finest foreground tower
    (models/networks.py:554-593 at ngf_s = 32: 64-byte pixels, half an LDS-DMA row) on the persistent single-chunk kernels
    (csrc/conv3x3_one_kernel.h) unchanged; the products that are not structural zeros are the layer's own, in the layer's own K order.
    Horizontal reflection in the paired domain is a CLAMP (pixel -1 = pixel 1 is the b = 1 half of paired pixel 0; the other half
    meets zero weights) -- the kernel's pair_x mode.  What PackedConv needs of an nn.Conv2d, assembled on read (include/v2v_hip.h,
    w_korder 3)."""

    def __init__(self, conv):
        if isinstance(conv, nn.ConvTranspose2d) or tuple(conv.kernel_size) != (3, 3) or tuple(conv.stride) != (1, 1) or conv.out_channels != 32 \
                or conv.in_channels > 32 or conv.groups != 1:
            raise ValueError("paired-x packing: 3x3 / stride 1 Conv2d, <= 32 -> 32 channels")
        self.src = conv
        self.is_transposed = False
        self.kernel_size, self.stride, self.padding, self.groups = conv.kernel_size, conv.stride, conv.padding, conv.groups
        self.in_channels, self.out_channels = 64, 64
        self.output_padding = (0, 0)

    def version_key(self):
        return _param_version(self.src)

    @property
    def weight(self):
        w = self.src.weight.detach().float()                     # [32][cin][3][3]
        cin = w.shape[1]
        wp = torch.zeros(2, 32, 2, 32, 3, 3, dtype=torch.float32, device=w.device)      # [a][co][b][ci][ky][kX]
        for a in range(2):
            for b in range(2):
                for kX in range(3):
                    kx = 2 * kX + b - a - 1
                    if 0 <= kx <= 2:
                        wp[a, :, b, :cin, :, kX] = w[:, :, :, kx]
        return wp.view(64, 64, 3, 3)

    @property
    def bias(self):
        return None if self.src.bias is None else torch.cat([self.src.bias.detach(), self.src.bias.detach()], 0)


class PairedXConvT:
    """The transposed counterpart of PairedXConv: a ConvTranspose2d(3x3, stride 2, padding 1, output_padding 1) with <= 32 input and
    exactly 16 output channels (models/networks.py:254-260 at ngf_s = 16: the finest foreground tower's last up-sampling stage), seen
    as a ConvTranspose2d 64 -> 32 over pairs of horizontally adjacent pixels: input [H][W][32] = [H][W/2][64], output [2H][2W][16] =
    [2H][W][32] (paired output pixel = output pixels 2x', 2x'+1; channel e*16+co), with
        W3[b*32+ci][e*16+co][ky][kx'] = W[ci][co][ky][kx],   kx = e+1-2b (kx' = 1),  3+e-2b (kx' = 2),  e-1-2b (kx' = 0)
    (zero where kx falls outside 0..2) -- the persistent transposed tile 114 unchanged."""

    def __init__(self, conv):
        if not isinstance(conv, nn.ConvTranspose2d) or tuple(conv.kernel_size) != (3, 3) or tuple(conv.stride) != (2, 2) \
                or tuple(conv.padding) != (1, 1) or tuple(conv.output_padding) != (1, 1) or conv.out_channels != 16 or conv.in_channels > 32 or conv.groups != 1:
            raise ValueError("paired-x packing: ConvTranspose2d(3x3, s2, p1, op1), <= 32 -> 16 channels")
        self.src = conv
        self.is_transposed = True
        self.kernel_size, self.stride, self.padding, self.groups = conv.kernel_size, conv.stride, conv.padding, conv.groups
        self.in_channels, self.out_channels = 64, 32
        self.output_padding = (1, 1)

    def version_key(self):
        return _param_version(self.src)

    @property
    def weight(self):
        w = self.src.weight.detach().float()                     # [cin][16][3][3]
        cin = w.shape[0]
        w3 = torch.zeros(2, 32, 2, 16, 3, 3, dtype=torch.float32, device=w.device)      # [b][ci][e][co][ky][kx']
        for b in range(2):
            for e in range(2):
                for kxp, kx in ((1, e + 1 - 2 * b), (2, 3 + e - 2 * b), (0, e - 1 - 2 * b)):
                    if 0 <= kx <= 2:
                        w3[b, :cin, e, :, :, kxp] = w[:, :, :, kx]
        return w3.view(64, 32, 3, 3)

    @property
    def bias(self):
        return None if self.src.bias is None else torch.cat([self.src.bias.detach(), self.src.bias.detach()], 0)


class PackedOneHot:
    """Gather table [49][cin][64 | 128] of a 7x7 stem Conv2d (csrc/onehot_stem.hip), refreshed like PackedConv."""

    def __init__(self, eng, mod, slice_, T, label_nc):
        self.mod = mod
        self.cout, self.cin = mod.weight.shape[0], mod.weight.shape[1]
        self.dtype, self.slice, self.T, self.label_nc = eng.dtype, slice_, T, label_nc
        n = lib.v2v_onehot_conv_table_bytes(self.cin, self.cout, self.dtype, slice_, T, label_nc)
        if n <= 0:
            raise RuntimeError("onehot_conv_table_bytes(%d, %d) failed" % (self.cin, self.cout))
        self.buf = torch.empty(n, dtype=torch.uint8, device=eng.device)
        self.bias = None
        self.version = None
        self.refresh()

    def refresh(self, force=False):
        w = self.mod.weight
        box = getattr(w, "_v2v_epoch", None)
        ver = (w._version, w.data_ptr(), None if self.mod.bias is None else self.mod.bias._version, _PARAM_EPOCH[0],
               0 if box is None else box[0])
        if not force and ver == self.version:
            return
        w32 = w.detach()
        if w32.dtype != torch.float32 or not w32.is_contiguous():
            w32 = w32.float().contiguous()
        check(lib.v2v_onehot_conv_pack_weights(_ptr(w32), _ptr(self.buf), self.cin, self.cout, self.dtype, self.slice,
                                               self.T, self.label_nc, _stream()),
              "onehot_conv_pack_weights")
        self.bias = None if self.mod.bias is None else self.mod.bias.detach().float().contiguous()
        self.version = ver


class Plan:
    """Recorded launch sequence (v2v_plan) with optional hipGraph replay."""

    def __init__(self):
        self.h = C.c_void_p(lib.v2v_plan_create())
        self.keep = []          # tensors the recorded launches point into
        self.graph_ready = False

    def __enter__(self):
        check(lib.v2v_plan_begin_record(self.h), "plan_begin_record")
        return self

    def __exit__(self, *exc):
        check(lib.v2v_plan_end_record(self.h), "plan_end_record")
        return False

    def label(self, text):
        lib.v2v_plan_set_label(self.h, text.encode())

    @property
    def num_ops(self):
        return lib.v2v_plan_num_ops(self.h)

    def run(self):
        check(lib.v2v_plan_run(self.h, _stream()), "plan_run")

    def instantiate_graph(self):
        check(lib.v2v_plan_instantiate_graph(self.h, _stream()), "plan_instantiate_graph")
        self.graph_ready = True

    def launch(self):
        if self.graph_ready:
            check(lib.v2v_plan_launch_graph(self.h, _stream()), "plan_launch_graph")
        else:
            self.run()

    def profile(self):
        n = self.num_ops
        ms = (C.c_float * n)()
        check(lib.v2v_plan_profile(self.h, _stream(), ms, n), "plan_profile")
        out = []
        for i in range(n):
            out.append((lib.v2v_plan_op_name(self.h, i).decode(), lib.v2v_plan_op_label(self.h, i).decode(), ms[i]))
        return out

    def timeline(self, graph=True):
        """[(op, label, lane, start_ms, end_ms)] of one replay with every lane on its own stream: captured as a hipGraph
        (the real schedule) or issued eagerly (shows the host's issue order as well)."""
        n = self.num_ops
        t0, t1, ln = (C.c_float * n)(), (C.c_float * n)(), (C.c_int32 * n)()
        fn = lib.v2v_plan_timeline_graph if graph else lib.v2v_plan_timeline
        check(fn(self.h, _stream(), t0, t1, ln, n), "plan_timeline")
        return [(lib.v2v_plan_op_name(self.h, i).decode(), lib.v2v_plan_op_label(self.h, i).decode(), int(ln[i]), t0[i], t1[i])
                for i in range(n)]

    def __del__(self):
        try:
            lib.v2v_plan_destroy(self.h)
        except Exception:
            pass


# tile config id -> (BM, BN, has a prefetch-helper instance)   (csrc/conv_igemm_kernel.h launch_typed)
TILE_CFGS = {1: (128, 128, False), 2: (128, 64, True), 3: (64, 64, True), 4: (128, 32, False), 5: (64, 128, True),
             6: (256, 64, False), 7: (128, 64, True), 8: (128, 128, False), 9: (64, 64, True), 10: (64, 64, False),
             11: (128, 64, True), 12: (64, 128, True), 13: (128, 64, True), 14: (128, 128, False),
             15: (128, 128, False), 16: (256, 64, False), 17: (64, 128, True), 18: (256, 128, False),
             19: (256, 128, False), 20: (128, 256, False), 21: (128, 128, False), 22: (256, 128, False),
             23: (128, 256, False)}
# LDS-resident-patch 3x3 kernel (csrc/conv3x3_patch_kernel.h): id -> (TH, TW, BN); weights in K order 1
PATCH_CFGS = {32: (2, 64, 64), 33: (4, 64, 64), 34: (2, 64, 128), 35: (4, 32, 64), 36: (8, 32, 64), 37: (4, 32, 128),
              40: (2, 64, 64), 41: (4, 64, 64), 42: (2, 64, 128), 43: (4, 32, 64), 44: (8, 32, 64), 45: (4, 32, 128),
              46: (4, 64, 64), 47: (2, 64, 128), 48: (4, 64, 128),   # 40+: dedicated loader waves
              # ping-pong wave groups (csrc/conv3x3_pp_kernel.h)
              50: (4, 64, 128), 51: (4, 64, 64), 52: (2, 64, 128), 53: (8, 32, 128), 54: (8, 32, 64), 55: (4, 32, 128),
              56: (8, 32, 64), 57: (4, 64, 64),
              # ping-pong, second schedule: LDS-DMA issued between the MFMAs; the only tiles v2v_conv2d_pair accepts
              # (csrc/conv3x3_pp2_kernel.h)
              70: (8, 32, 64), 71: (8, 32, 128), 72: (8, 32, 64), 73: (4, 64, 64), 74: (4, 64, 64), 75: (4, 32, 128),
              # single-phase software-pipelined schedule (csrc/conv3x3_pp3_kernel.h): two fragment register sets, ONE barrier per step
              80: (8, 32, 64), 81: (8, 32, 128), 82: (8, 32, 64), 83: (4, 64, 64), 84: (4, 32, 128), 85: (4, 64, 128),
              86: (4, 32, 128), 87: (2, 64, 128),      # 86 / 87: four waves, 64 x 64 wave tiles
              90: (8, 32, 64), 91: (4, 64, 64),        # 90 / 91: 82 / 83 with K pairs (8 fragment reads per 8 MFMAs)
              92: (8, 32, 64), 93: (4, 64, 64),        # 92 / 93: K quads (6 reads per 8 MFMAs)
              94: (8, 32, 64), 95: (4, 64, 64), 96: (4, 32, 64),
              # (the round-5 experiment tiles 97-99 / 130-132 on tile 90's geometry and 142 -- all bit-identical to the tiles they varied, none
              #  faster, DESIGN 3.1 -- were removed in round 6)
              # csrc/conv3x3_one_kernel.h (round 5): PERSISTENT, weights-resident tile for single-chunk layers with <= 64 output channels
              # (one workgroup per CU walks its tiles; no barrier / DMA / wait inside a tile's 9 steps).  140: bit-identical to tile 94 and
              # 18-20 % faster (profiles/r05_v2_one_bench.txt, r05_v3_one_bench.txt); 141 (two patch buffers): another 5 % on the 2048-tile
              # layer, 5-7 % slower on the small ones (profiles/r05_v6_stagger.txt) -- both are offered, the search decides per shape;
              # 143: 141 with its stores left in flight (experiment, V2V_EXP_TILES=1).   94 / 95: single-chunk layers (64 bf16 input
              # channels): one patch buffer, 3 weight stages (72 / 80 KiB)
              140: (8, 32, 64), 141: (8, 32, 64), 143: (8, 32, 64)}
# stride-2 3x3 convolutions on the plane-resident patch kernel (csrc/conv3x3_s2_kernel.h): id -> (TH, TW, BN) of the OUTPUT tile
S2_CFGS = {100: (4, 32, 64), 101: (4, 32, 128), 102: (4, 32, 64), 103: (4, 32, 128)}
# ConvTranspose2d(3x3, stride 2) with all four output-parity classes per workgroup (csrc/conv3x3_t2_kernel.h): id -> (TH, TW, BN), tile of INPUT positions
T2_CFGS = {110: (4, 32, 64), 111: (4, 32, 128), 112: (8, 32, 64), 113: (4, 32, 64),
           114: (8, 32, 32)}     # 114: persistent, weights resident, single chunk (64 input channels), <= 32 output channels (csrc/conv3x3_one_kernel.h)
# dense 7x7 / stride 1 / pad 3 convolutions on the single-phase kernel with a 7x7 window (csrc/conv3x3_pp3_kernel.h, KK = 7): id -> (TH, TW, BN).
# Round 5: validated on the GPU (tests/test_gpu_kernels.py::test_conv7x7_window_tiles), offered to the tile search unless V2V_S7_PATCH=0
S7_CFGS = {120: (4, 32, 64), 121: (4, 32, 128)}
ABLATION_TILES = {78: (8, 32, 128), 79: (8, 32, 64), 88: (8, 32, 128), 89: (8, 32, 64)}     # instrumented copies of 71 / 70 (scripts/pp2_ablate.py); never auto-selected
PAIR_TILES = (70, 71, 72, 73, 74, 75, 80, 81, 82, 83, 84, 85, 86, 87, 90, 91, 92, 93)
ONE_TILES = (140, 141, 143)                                     # persistent, weights-resident single-chunk tiles (csrc/conv3x3_one_kernel.h)
PERSISTENT_TILES = ONE_TILES + (114,)                           # ... and the transposed stride-2 one: ONE statistics row per workgroup, finalize in the launch at any size
ONE_FIN = os.environ.get("V2V_ONE_FIN", "1") != "0"             # ... finalize their <= 256 statistics rows in the launch (0: separate bn_finalize launch, for A/B)
EXP_TILES = (143,)                                              # (140 / 141: validated and faster -- regular tiles since visits r05_v3 / r05_v6)
if os.environ.get("V2V_EXP_TILES", "0") == "1":
    PAIR_TILES = PAIR_TILES + EXP_TILES


def is_patch_tile(t):
    """Tile ids of the LDS-patch 3x3 kernels (weights in K order 1): patch 32-48, ping-pong 50-57, ping-pong 2 70-79, single-phase
    80-99, its round-5 experiment tiles 130-139 and the persistent single-chunk tile 140; stride-2 100-109, 7x7 window 120-129."""
    return 32 <= t < 60 or 70 <= t < 110 or 120 <= t < 150


def tile_korder(t):
    """Weight packing a tile id reads: 0 tap-major class matrices (implicit-GEMM tiles, 7x7 kernels), 1 channel-chunk-major (patch
    kernels, stride-2 patch kernel), 2 the full-tap chunk-major matrix of a transposed layer (conv3x3_t2_kernel, ids 110-119)."""
    return 2 if 110 <= t < 120 else 1 if is_patch_tile(t) else 0
FUSE_FINALIZE_MAX_PIXELS = 32768   # larger layers leave thousands of statistics rows: parallel two-stage finalize instead
PREFETCH_DIST = 12          # K chunks (128 B of every weight row each) the helper wave runs ahead


def _cfg3(v):
    """tile_override / _tuned values: tile id, or (tile, splitk, prefetch)."""
    if isinstance(v, (tuple, list)):
        return int(v[0]), int(v[1]), int(v[2])
    return int(v), 1, 0


# Ticket words of one (lane, scratch set): [0, 256) the conv kernels' per-channel-tile arrive / depart tickets, [256, 256 + 64 * 128)
# the row-group tickets of the two-level in-kernel finalize (conv_igemm_kernel.h: fin_counter + 256 + g * n_tiles + nt), then the
# one-hot stems' per-slice tickets (onehot_stem.hip: fin_counter + blockIdx.y) in a region of their own (they used to
# start at word 256, inside the row-group region -- safe only while launches of one lane are serialised).
FIN_ONEHOT_OFFSET = 256 + 64 * 128
FIN_TAG_OFFSET = FIN_ONEHOT_OFFSET + 256            # include/v2v_hip.h V2V_FIN_TAG_WORD: the fused-norm launches' per-channel-tile launch tags
FIN_COUNTER_WORDS = FIN_TAG_OFFSET + 128


class Engine:
    """Emits v2v_* launches for one device / one activation dtype."""

    def __init__(self, device, dtype=L.F32, align_corners=False, record_only=False, x3=False):
        self.device = torch.device(device)
        # x3 (fp32 engine only): 3x3 convolutions whose input channels are whole 128-byte chunks run on the bf16 matrix pipe
        # over bf16x3 operands (X3Conv / split_x3): fp32-grade products at 3x the bf16 cost instead of the fp32-input MFMA
        # rate; statistics, norms, activations and everything else stay fp32
        self.x3 = bool(x3) and dtype == L.F32
        self._x3_engine = None
        self._x3_convs = {}
        # record_only: launches may only be RECORDED into a Plan (never executed).  Used by the CPU
        # test-suite to check the lowering (layer census, argument validation) without a GPU.
        self.record_only = record_only
        if self.device.type != "cuda" and not record_only:
            raise RuntimeError("vid2vid_amd runs on MI355X only (got device %s); there is no CPU path" % device)
        if record_only:
            # dry run: every v2v_* call still validates its arguments, nothing is ever launched (include/v2v_hip.h,
            # v2v_set_dry_run) -- lets the CPU test-suite drive whole training / inference control flows
            lib.v2v_set_dry_run(1)
        elif lib.v2v_get_dry_run():
            # the dry-run switch is process-wide: with it set, every launch, plan run and optimizer step of THIS engine
            # would return success without executing anything -- refuse instead of computing nothing
            raise RuntimeError("v2v dry-run mode is on (a record-only Engine was created in this process): call "
                               "vid2vid_amd.networks.set_record_only(False) before creating a GPU engine")
        self.dtype = dtype
        self.tdtype = _TORCH_DTYPE[dtype]
        self.align_corners = align_corners
        self._packed = {}        # id(module) -> PackedConv
        self._packed_onehot = {} # id(module) -> PackedOneHot (7x7 stems fed by label maps)
        # stems over encoded label maps as a weight gather-sum (csrc/onehot_stem.hip); V2V_ONEHOT_STEM=0: dense conv on the encoding
        self.onehot_stem = bool(int(os.environ.get("V2V_ONEHOT_STEM", "1")))
        self.onehot_slice = int(os.environ.get("V2V_ONEHOT_SLICE", "0"))      # output channels per workgroup: 32 / 64, 0 = default
        # conv + norm + activation + residuals in one launch (spin barrier between the workgroups of a channel tile) for
        # the paired ResnetBlock convolutions; V2V_FUSED_NORM=0: raw fp32 output + bn_apply launch
        self.fused_norm = bool(int(os.environ.get("V2V_FUSED_NORM", "1")))
        # V2V_RAW_BF16=1 (bf16 storage, inference plans): the pre-norm output of an UNPAIRED convolution (stride-2 / transposed stages,
        # the foreground tower, fine-scale ResnetBlocks) is stored as bf16 instead of fp32 (V2V_OUT_RAW_ACT_NHWC) -- the conv writes and
        # the bn_apply reads half the bytes; the statistics still come from the fp32 accumulators.  Built, tested
        # (test_conv_raw_output_in_activation_dtype) and measured on one box (profiles/r04_a6_raw_bf16_ab.txt): bn_apply -7 % / -10 %,
        # but the convolutions' epilogues pay more for the bf16 packing than the stores save (+3 % / +7 % conv time) -- the frame is
        # 1.7 % SLOWER at both resolutions and the bf16 error grows (fake_B mean 1.44e-2 -> 1.64e-2).  OFF by default.
        self.raw_bf16 = bool(int(os.environ.get("V2V_RAW_BF16", "0")))
        self.raw_bf16_one = os.environ.get("V2V_RAW_BF16_ONE", "1") != "0"      # bf16 raw output of the persistent single-chunk tiles (Engine.conv)
        # V2V_HEAD_ROWSUM=0: the bf16 generator heads (7x7, <= 4 output channels, planar fp32) stay on conv7x7_head_kernel (tile 60)
        # instead of conv7x7_rowsum_kernel (tile 62: row GEMM over (kernel column, channel) + shifted sum) -- A/B switch
        self.rowsum_heads = bool(int(os.environ.get("V2V_HEAD_ROWSUM", "1")))
        self._fused_norm_wgs = None
        # model_final_flow + model_final_w (same input) as one 7x7 head launch; V2V_MERGE_HEADS=0: one launch each
        self.merge_heads = bool(int(os.environ.get("V2V_MERGE_HEADS", "1")))
        self._merged_heads = {}
        self._scratch = {}       # name -> tensor (grown on demand, shared between layers)
        self._grids = {}
        self._zero_page = None
        self._fin_counters = {}   # per plan lane: concurrent branches must not share ticket words
        self._sk_counters = {}
        self._lane = 0           # current plan lane (hipGraph branch); selects the scratch set
        self._sset = 0           # scratch sub-set inside a lane: member 1 of a paired launch owns its own raw / stats / tickets
        self.twin_enabled = False    # set by the frame plan: twin chains (label / image towers, image / flow branches) as paired launches
        self.pair_override = None    # (tile, splitk) forced for every paired launch (experiments: V2V_PAIR_TILE="70,1")
        if os.environ.get("V2V_PAIR_TILE"):
            self.pair_override = tuple(int(v) for v in os.environ["V2V_PAIR_TILE"].split(","))
        self.lanes_enabled = False   # set by the frame plan: emit independent towers / branches on parallel lanes
        self._thrash = None
        # Round 6: weight-gradient launches of a backward pass go to a SIDE stream (autograd.ConvFn.backward).  dW is a leaf of
        # the backward graph -- nothing downstream of a layer's backward needs it before the optimizer step -- while the chain
        # norm-backward -> backward-data -> next layer is strictly serial and leaves the chip idle between its ~4000 dependent
        # launches per chunk (the device is busy 70 % of a chunk, and replaying the chunk as ONE hipGraph does not change that:
        # the gaps are the device's kernel-to-kernel turnaround, not host time; profiles/r06_v3_traingraph.txt).  V2V_WGRAD_STREAM=0
        # keeps everything on one stream.
        self.wgrad_stream_on = os.environ.get("V2V_WGRAD_STREAM", "1") != "0"
        self._wgrad_stream = None
        self._wgrad_join_queued = False
        self.repack_async_on = os.environ.get("V2V_REPACK_ASYNC", "1") != "0"
        self._repack_event = None
        import weakref
        _ENGINES.append(weakref.ref(self))
        self.plan = None        # Plan being recorded (for labels / keep-alive)
        self.conv_log = []       # (label, desc summary) of every conv emitted; used by bench/roofline
        self.tile_override = {}  # (cin,cout,KH,stride,transposed) -> tile id (tests / manual tuning)
        # V2V_TILE_OVERRIDE="cin,cout,K,stride,transposed:tile,splitk,prefetch;..." pins configurations (experiments)
        for item in os.environ.get("V2V_TILE_OVERRIDE", "").split(";"):
            if ":" in item:
                k, v = item.split(":")
                self.tile_override[tuple(int(x) for x in k.split(","))] = tuple(int(x) for x in v.split(","))
        self.autotune = False    # measure the tile configurations once per conv shape (plan build time)
        self._tuned = {}         # (cin,cout,KH,stride,transposed,N,H,W,out_mode,Cs) -> (tile, splitk, prefetch)
        self.bwd_tile_override = {}   # (dY channels, dX channels, KH, stride, transposed) of a backward-data operator -> tile (tests)
        self._tune_alts = {}     # same key -> runner-up configurations of the isolated search (this process only)
        self._tune_wide = {}     # same key -> the wider candidate list used for the heaviest shapes of a frame
        # optional persistent tuning cache (V2V_TUNE_CACHE=<json>): a profiling run can replay exactly the
        # configurations a previous benchmark run selected instead of re-measuring them under the profiler
        self._tune_cache_path = os.environ.get("V2V_TUNE_CACHE", "")
        if self._tune_cache_path and os.path.exists(self._tune_cache_path):
            try:
                with open(self._tune_cache_path) as f:
                    for k, v in json.load(f).get(str(dtype), {}).items():
                        self._tuned[tuple(int(x) for x in k.split(","))] = tuple(v)
            except (OSError, ValueError):
                pass
        self.update_running_stats = False
        self.fused_finalize = True   # norm statistics finalized by the conv kernel's last workgroup (small layers)
        # round 4, V2V_FIN2=1: layers with more than 512 statistics rows (> 32768 output pixels) finalize in the conv launch too, in
        # two levels (v2v_conv_desc.fin_workspace; bit for bit v2v_bn_finalize: test_two_level_in_kernel_finalize_equals_bn_finalize).
        # Built, tested, measured on one box (profiles/r04_b7_fin2_ab.txt): 66 launches fewer per 2048x1024 frame (355 -> 289), the
        # two finalize kernels -0.45 ms, but every workgroup of those thousand-tile launches now drains its stores and takes a ticket
        # before it exits: conv +1.0 ms -- 83.1 -> 80.3 frames/s (512x256: 372 -> 367).  OFF by default.
        self.fused_finalize2 = bool(int(os.environ.get("V2V_FIN2", "0")))
        self.last_finalized = False  # did the last conv() finalize its statistics in-kernel?
        self.ablate = 0              # profiling ablations (scripts/conv_ablate.py); results are wrong when set

    # ---------------- buffers ----------------
    def empty_act(self, N, H, W, C):
        t = torch.empty((N, H, W, pad_channels(C, self.dtype)), dtype=self.tdtype, device=self.device)
        self._keep(t)
        return Act(t, C)

    def zeros_act(self, N, H, W, C):
        a = self.empty_act(N, H, W, C)
        a.t.zero_()
        return a

    def empty_f32(self, *shape):
        t = torch.empty(shape, dtype=torch.float32, device=self.device)
        self._keep(t)
        return t

    def scratch(self, name, numel, dtype=torch.float32, zero=False):
        """Shared scratch (conv raw output, statistics): consumed by the next launch on the
        same stream, so one buffer per kind is enough.  zero: the buffer starts as zeros (every time it is (re)allocated)."""
        key = (name, dtype, self._lane, self._sset)
        t = self._scratch.get(key)
        if t is None or t.numel() < numel:
            if self.plan is not None and t is not None and not self.record_only:
                raise RuntimeError("scratch '%s' must not grow while a plan is recording" % name)
            t = (torch.zeros if zero else torch.empty)(int(numel), dtype=dtype, device=self.device)
            self._scratch[key] = t
        self._keep(t)
        return t

    def reserve_scratch(self, raw_elems, stats_elems):
        self.scratch("raw", raw_elems)
        self.scratch("stats", stats_elems)

    def _keep(self, t):
        if self.plan is not None:
            self.plan.keep.append(t)

    def label(self, text):
        if self.plan is not None:
            self.plan.label(text)

    @property
    def _sk_counter(self):       # lane-0 split-K ticket words (tests check that kernels re-arm them)
        return self._sk_counters.get((0, 0))

    @property
    def _fin_counter(self):
        return self._fin_counters.get((0, 0))

    @contextlib.contextmanager
    def scratch_set(self, k):
        """Scratch sub-set k of the current lane (raw conv output, statistics, tickets, split-K slabs): the second member
        of a paired launch must not share them with the first."""
        prev, self._sset = self._sset, k
        try:
            yield
        finally:
            self._sset = prev

    # ---------------- plan lanes (parallel hipGraph branches) ----------------
    @contextlib.contextmanager
    def on_lane(self, k):
        """Emit the enclosed launches on plan lane k: a branch that starts after everything emitted so far on the
        current lane and runs concurrently with what the current lane emits next (include/v2v_hip.h, v2v_plan_set_lane).
        Each lane has its own scratch set / ticket words.  Without `lanes_enabled` this is a no-op."""
        parent = self._lane
        if not self.lanes_enabled or k == parent:
            yield
            return
        check(lib.v2v_plan_lane_wait(k, parent), "plan_lane_wait")       # fork
        self._lane = k
        check(lib.v2v_plan_set_lane(k), "plan_set_lane")
        try:
            yield
        finally:
            self._lane = parent
            check(lib.v2v_plan_set_lane(parent), "plan_set_lane")

    def join(self, k):
        """The current lane continues after everything emitted so far on lane k."""
        if self.lanes_enabled and k != self._lane:
            check(lib.v2v_plan_lane_wait(self._lane, k), "plan_lane_wait")

    def wgrad_side_stream(self):
        """The stream weight-gradient kernels run on, or None (switch off / no device / record-only engine)."""
        if not self.wgrad_stream_on or self.record_only or self.device.type != "cuda":
            return None
        if self._wgrad_stream is None:
            self._wgrad_stream = torch.cuda.Stream(device=self.device)
            from . import parallel
            parallel.register_grad_stream(self._wgrad_stream)      # bucket all-reduces sent from inside the pass must wait for it too
        return self._wgrad_stream

    def queue_wgrad_join(self):
        """Called from inside a backward pass after the first side-stream launch: when the autograd engine has finished the pass,
        the stream that called backward() waits for the side stream -- whatever reads .grad next (optimizer step, a test, a
        gradient clip) is ordered behind every weight-gradient kernel without knowing about the side stream."""
        if self._wgrad_join_queued:
            return
        self._wgrad_join_queued = True
        side = self._wgrad_stream

        def join():
            self._wgrad_join_queued = False
            torch.cuda.current_stream(self.device).wait_stream(side)
        torch.autograd.Variable._execution_engine.queue_callback(join)

    def zero_page(self):
        """256 zero bytes: source of padded / ragged lanes of the LDS-DMA loaders."""
        if self._zero_page is None:
            self._zero_page = torch.zeros(256, dtype=torch.uint8, device=self.device)
        return self._zero_page

    def grid(self, H, W):
        """Base sampling grid of get_grid (models/networks.py:79-93): torch.linspace(-1,1,n)."""
        key = (H, W)
        if key not in self._grids:
            gx = torch.linspace(-1.0, 1.0, W).to(self.device)
            gy = torch.linspace(-1.0, 1.0, H).to(self.device)
            self._grids[key] = (gx, gy)
        return self._grids[key]

    # ---------------- weights ----------------
    def packed(self, mod, cin_stride, role="fwd", reflect=False, korder=0, refresh=True, touch=True):
        """touch=False: the caller only wants the layer's geometry -- the packing is not marked as in use (repack_async re-packs
        after every optimizer step what was used since the last one: a patch-tile layer's tap-major packing, fetched by Engine.conv
        for its geometry and never read by a launch, was a third of those re-packs)."""
        key = (id(mod), cin_stride, role, reflect, korder)
        if self._repack_event is not None:
            self.wait_repack()    # packings refreshed on the side stream after the last optimizer step: this stream reads them from here on
        pc = self._packed.get(key)
        if pc is None:
            pc = PackedConv(self, mod, cin_stride, role, reflect, korder)
            self._packed[key] = pc
            pc.used = touch
        elif self.plan is None and refresh:
            pc.refresh()          # eager (training) use: follow optimizer updates
        if touch:
            pc.used = True
        return pc

    def repack_async(self, flat=None):
        """Refresh the stale packings of `flat`'s parameters that were used since the last call, on the side stream (see
        repack_after_step).  Packings whose owner is unknown (parameters re-homed after the packing was made, composed views) are
        left to the lazy refresh in front of their next use."""
        side = self.wgrad_side_stream()
        if side is None or self.plan is not None or not self.repack_async_on or torch.cuda.is_current_stream_capturing():
            return
        fid = id(flat) if flat is not None else None
        todo = [pc for pc in self._packed.values() if pc.used and (fid is None or pc.flat_id == fid)]
        if not todo:
            return
        cur = torch.cuda.current_stream(self.device)
        side.wait_stream(cur)                       # behind the optimizer kernel (and every reader of the old packings)
        with torch.cuda.stream(side):
            for pc in todo:
                pc.used = False
                pc.refresh()
            ev = torch.cuda.Event()
            ev.record(side)
        self._repack_event = ev

    def wait_repack(self):
        ev = self._repack_event
        if ev is not None:
            self._repack_event = None
            torch.cuda.current_stream(self.device).wait_event(ev)

    def refresh_weights(self, force=False):
        for pc in self._packed.values():
            pc.refresh(force)
        for pc in self._packed_onehot.values():
            pc.refresh(force)
        if self._x3_engine is not None:                 # the bf16x3 weight views live in the sub-engine (X3Conv.version_key follows the source layer)
            self._x3_engine.refresh_weights(force)

    # ---------------- one-hot stem (label-map input) ----------------
    def onehot_conv_ok(self, conv, cin, H, W, pad_mode=L.PAD_REFLECT, pad=3):
        """A 7x7 / stride 1 / ReflectionPad2d(3) Conv2d that can run as the gather-sum on label maps (inference only)."""
        return (self.onehot_stem and isinstance(conv, nn.Conv2d) and conv.kernel_size == (7, 7)
                and conv.stride == (1, 1) and conv.groups == 1 and pad_mode == L.PAD_REFLECT and pad == 3
                and conv.out_channels <= 128 and conv.in_channels == cin and H >= 4 and W >= 4
                and not (self.plan is None and torch.is_grad_enabled()))

    def onehot_eligible(self, x, conv, pad_mode, pad_override):
        """... straight on an encoded label Act."""
        return (x.onehot is not None and x.N == 1
                and self.onehot_conv_ok(conv, x.C, x.H, x.W, pad_mode, conv.padding[0] if pad_override is None else pad_override))

    def label_codes(self, src, H, W):
        """One byte per (frame, pixel): label | instance-edge << 7 (v2v_label_codes), shared by every stem of the frame."""
        if src.label_nc > 126:
            return None
        codes = torch.empty((src.T, H, W), dtype=torch.uint8, device=self.device)
        self._keep(codes)
        check(lib.v2v_label_codes(_ptr(src.labels), _ptr(src.inst), int(src.labels.dtype == torch.uint8), _ptr(codes), src.T, H, W,
                                  src.label_nc, _stream()), "label_codes")
        self.label("label_codes")
        src.codes = codes
        return codes

    def onehot_conv(self, x, conv, want_stats=True, label="", fin=None):
        """Raw fp32 NHWC output + statistics rows of the stem convolution, computed from the label maps behind `x`.
        Returns (raw, rows, (N, OH, OW)) like conv(..., OUT_RAW_F32_NHWC, want_stats=True).  fin = (norm, ss): the
        statistics are finalized in the kernel (last workgroup of each channel slice), no bn_finalize launch."""
        src = x.onehot
        pk = self._packed_onehot.get((id(conv), self.onehot_slice))
        if pk is None:
            pk = self._packed_onehot[(id(conv), self.onehot_slice)] = PackedOneHot(self, conv, self.onehot_slice, src.T, src.label_nc)
        elif self.plan is None:
            pk.refresh()
        H, W, cout = x.H, x.W, conv.out_channels
        cs = (cout + 3) // 4 * 4
        raw = self.scratch("raw", H * W * cs)
        rows = lib.v2v_onehot_conv_stats_rows(H, W) if want_stats else 0
        st = self.scratch("stats", rows * cout * 2) if want_stats else None
        self._keep(pk.buf)
        if pk.bias is not None:
            self._keep(pk.bias)
        if src.codes is not None:
            lab_ptr, in_mode = _ptr(src.codes), 2
        else:
            lab_ptr, in_mode = _ptr(src.labels), int(src.labels.dtype == torch.uint8)
        if fin is not None and want_stats:
            norm, ss = fin
            gamma, beta, eps, mom, rm, rv = self._norm_params(norm, 1)
            key = (self._lane, self._sset)
            fin_counter = self._fin_counters.get(key)
            if fin_counter is None:
                fin_counter = self._fin_counters[key] = torch.zeros(FIN_COUNTER_WORDS, dtype=torch.int32, device=self.device)
            fn = L.OneHotNorm()
            fn.counter = fin_counter.data_ptr() + 4 * FIN_ONEHOT_OFFSET  # own region: clear of the conv kernels' per-tile tickets (0..255) AND of the two-level finalize's row-group tickets (256 .. 256 + 64*128)
            fn.gamma = None if gamma is None else gamma.data_ptr()
            fn.beta = None if beta is None else beta.data_ptr()
            fn.scale_shift = ss.data_ptr()
            fn.running_mean = None if rm is None else rm.data_ptr()
            fn.running_var = None if rv is None else rv.data_ptr()
            fn.eps, fn.momentum, fn.count = eps, mom, H * W
            for t in (gamma, beta, ss, fin_counter):
                if t is not None:
                    self._keep(t)
            check(lib.v2v_onehot_conv7x7_norm(lab_ptr, _ptr(src.inst), in_mode, _ptr(pk.buf),
                                              _ptr(pk.bias), _ptr(raw), _ptr(st), src.T, H, W, src.label_nc, cout, cs, self.dtype,
                                              pk.slice, C.byref(fn), _stream()), "onehot_conv7x7_norm " + label)
        else:
            check(lib.v2v_onehot_conv7x7(lab_ptr, _ptr(src.inst), in_mode, _ptr(pk.buf),
                                         _ptr(pk.bias), _ptr(raw), _ptr(st), src.T, H, W, src.label_nc, cout, cs, self.dtype,
                                         pk.slice, _stream()), "onehot_conv7x7 " + label)
        self.label(label)
        if len(self.conv_log) >= 100000:
            del self.conv_log[:]
        # flops: those of the dense convolution this launch replaces (the frame's nominal 2115 GFLOP census), flagged
        self.conv_log.append(dict(label=label, N=1, H=H, W=W, OH=H, OW=W, cin=conv.in_channels, cout=cout, tune_key=None,
                                  KH=7, KW=7, stride=1, transposed=False, onehot=True,
                                  flops=2.0 * H * W * cout * conv.in_channels * 49, tile=(0, 0), splitk=1, prefetch=0))
        return raw, rows, (1, H, W)

    # ---------------- primitive emitters ----------------
    def conv(self, x, mod, pad_mode=L.PAD_ZERO, pad_override=None, out_mode=L.OUT_RAW_F32_NHWC,
             act=L.ACT_NONE, act_param=0.0, out_scale=1.0, want_stats=False, out=None, label="", fin=None, act_b=None, raw_act_ok=False):
        """Emit one convolution.  Returns (out, stats_rows, (N,OH,OW)).
        fin = (norm module, ss tensor [4*cout]): finalize the training-mode norm statistics inside the conv
        kernel (last-arriving workgroup), so no separate bn_finalize launch is needed."""
        pc = self.packed(mod, x.Cs, refresh=False, touch=False)     # geometry now; the packing that the chosen tile reads is fetched (and refreshed) below
        pad = pc.pad if pad_override is None else pad_override
        N, H, W = x.N, x.H, x.W
        if pc.transposed:
            OH = (H - 1) * pc.stride - 2 * pad + pc.KH + pc.out_pad
            OW = (W - 1) * pc.stride - 2 * pad + pc.KW + pc.out_pad
        else:
            OH = (H + 2 * pad - pc.KH) // pc.stride + 1
            OW = (W + 2 * pad - pc.KW) // pc.stride + 1
        d = ConvDesc()
        d.in_ = x.t.data_ptr(); d.w = pc.buf.data_ptr(); d.w_korder = 0
        d.zero_page = self.zero_page().data_ptr()
        self._keep(self._zero_page)
        d.bias = None if pc.bias is None else pc.bias.data_ptr()
        d.N, d.H, d.W = N, H, W
        d.cin, d.cin_stride, d.cout = pc.cin, x.Cs, pc.cout
        d.KH, d.KW, d.stride, d.pad, d.pad_mode = pc.KH, pc.KW, pc.stride, pad, pad_mode
        d.transposed = int(pc.transposed)
        d.OH, d.OW = OH, OW
        # raw_act_ok (conv_group -> norm_apply only): the raw tensor may be stored in the activation dtype.  Never on the autograd
        # path (the backward kernels read fp32 raw), never for the 7x7 layers (tiles 60 / 61 and the gather-sum stems write fp32)
        raw_ok_t = bool(raw_act_ok and out_mode == L.OUT_RAW_F32_NHWC and self.dtype == L.BF16 and out is None
                        and pc.KH != 7 and not torch.is_grad_enabled())
        raw_t = raw_ok_t and self.raw_bf16
        d.dtype, d.out_mode, d.act = self.dtype, (L.OUT_RAW_ACT_NHWC if raw_t else out_mode), act
        d.act_param, d.out_scale = act_param, out_scale
        if act_b is not None:          # (first channel, activation, parameter, scale) of the second head of a merged pair
            d.act_split, d.act_b, d.act_param_b, d.out_scale_b = act_b
        d.ablate = self.ablate
        d.tile, d.splitk, d.prefetch = _cfg3(self.tile_override.get((pc.cin, pc.cout, pc.KH, pc.stride, int(pc.transposed)), 0))
        tune_key = (pc.cin, pc.cout, pc.KH, pc.stride, int(pc.transposed), N, H, W, out_mode, x.Cs)
        if d.tile == 0 and tune_key in self._tuned:
            d.tile, d.splitk, d.prefetch = _cfg3(self._tuned[tune_key])
        if raw_ok_t and not raw_t and self.raw_bf16_one and d.tile in PERSISTENT_TILES:
            # round 6: the persistent single-chunk tiles write their raw output rounded to bf16 (the statistics come from the fp32
            # accumulators): these layers are HBM-bound and the raw tensor is written once and read once by bn_apply -- 4 instead of
            # 8 bytes per element for the pair.  V2V_RAW_BF16_ONE=0 keeps fp32 raw.
            raw_t = True
            d.out_mode = L.OUT_RAW_ACT_NHWC
        if d.tile in PERSISTENT_TILES and not (self.dtype == L.BF16 and d.out_mode in (L.OUT_RAW_F32_NHWC, L.OUT_RAW_ACT_NHWC)):
            # a cached / overridden selection made under another raw-output mode or dtype (the key holds the LOGICAL out_mode:
            # V2V_RAW_BF16 flips the effective one; an fp32 engine may share the cache file): the persistent tiles write fp32 raw
            # from bf16 only -- fall back to the library's default tile instead of failing at launch
            d.tile, d.splitk, d.prefetch = 0, 1, 0
        if act_b is not None:
            d.tile, d.splitk, d.prefetch = 60, 1, 0          # per-channel epilogues exist in the 7x7 head kernels only
        if d.tile == 60 and self.rowsum_heads and (pc.cin, pc.cout, pc.KH, pc.stride, int(pc.transposed)) not in self.tile_override \
                and self.dtype == L.BF16 and pc.cout <= 4 and out_mode == L.OUT_F32_NCHW \
                and x.Cs % 32 == 0 and not want_stats and fin is None:
            d.tile = 62                                       # conv7x7_rowsum_kernel: the heads as row GEMM + shifted sum (bf16)
        if tile_korder(d.tile):
            pc = self._use_korder(d, mod, x.Cs, tile_korder(d.tile))
        else:
            pc.used = True             # this launch reads the tap-major packing
            if self.plan is None:
                pc.refresh()           # eager use: follow optimizer updates (only the packing this launch reads)
        d.bias = None if pc.bias is None else pc.bias.data_ptr()      # of the packing in use (aliases the parameter)
        if pc.cin != x.C:
            raise RuntimeError("conv %s: input has %d channels, layer expects %d" % (label, x.C, pc.cin))
        if raw_t:
            cs = (pc.cout + 7) // 8 * 8
            d.cout_stride = cs
            out = self.scratch("raw", (N * OH * OW * cs + 1) // 2)[:N * OH * OW * cs // 2].view(torch.bfloat16)     # the same scratch, half the bytes
            d.out = out.data_ptr()
        elif out_mode == L.OUT_RAW_F32_NHWC:
            cs = (pc.cout + 3) // 4 * 4
            d.cout_stride = cs
            if out is None:
                out = self.scratch("raw", N * OH * OW * cs)
            d.out = out.data_ptr()
        elif out_mode == L.OUT_ACT_NHWC:
            if out is None:
                out = self.empty_act(N, OH, OW, pc.cout)
                if out.Cs != pc.cout:
                    out.t.zero_()          # padded channels must read as zero downstream
            d.cout_stride = out.Cs
            d.out = out.t.data_ptr()
        else:
            if out is None:
                out = self.empty_f32(N, pc.cout, OH, OW)
            d.cout_stride = pc.cout
            d.out = out.data_ptr()
        rows = 0
        if want_stats:
            d.stats = None
            rows = lib.v2v_conv_stats_rows(C.byref(d))
            if rows <= 0:
                check(rows or -1, "conv_stats_rows")
            st = self.scratch("stats", rows * pc.cout * 2)
            d.stats = st.data_ptr()
            # (not for the 7x7 layers: their halo-patch tiles 60 / 61 have no in-kernel finalize and must stay eligible)
            two_level = bool(fin is not None and N * OH * OW > FUSE_FINALIZE_MAX_PIXELS and self.fused_finalize2 and rows > 512
                             and pc.KH != 7 and d.tile not in (60, 61))
            if fin is not None and N * OH * OW > FUSE_FINALIZE_MAX_PIXELS and not two_level and not (d.tile in PERSISTENT_TILES and ONE_FIN):
                fin = None             # one workgroup walking >1000 rows costs 0.1-1 ms (profiles/r01_v15_finalize_tail.txt)
                                       # (the persistent tiles leave one row per WORKGROUP, <= 256: they finalize in the launch at any size)
            self.last_finalized = fin is not None
            if fin is not None:
                norm, ss = fin
                gamma, beta, eps, mom, rm, rv = self._norm_params(norm, N)
                fin_counter = self._fin_counters.get((self._lane, self._sset))
                if fin_counter is None:
                    # [0,128) finalize tickets per channel tile, [128,256) fused-norm departures, [256, 256 + 64 * 128) row-group tickets
                    fin_counter = self._fin_counters[(self._lane, self._sset)] = torch.zeros(FIN_COUNTER_WORDS, dtype=torch.int32, device=self.device)
                d.fin_counter = fin_counter.data_ptr()
                d.fin_gamma = None if gamma is None else gamma.data_ptr()
                d.fin_beta = None if beta is None else beta.data_ptr()
                d.fin_scale_shift = ss.data_ptr()
                d.fin_running_mean = None if rm is None else rm.data_ptr()
                d.fin_running_var = None if rv is None else rv.data_ptr()
                d.fin_eps, d.fin_momentum, d.fin_count = eps, mom, N * OH * OW
                if two_level:          # round 4: large layers finalize inside the conv launch in two levels (no bn_partial_reduce / bn_finalize launches)
                    ws = self.scratch("bn_ws", 64 * pc.cout * 2, torch.float64)
                    d.fin_workspace = ws.data_ptr()
                for t in (gamma, beta, ss, fin_counter):
                    if t is not None:
                        self._keep(t)
        if pc.bias is not None:
            self._keep(pc.bias)
        if (self.autotune and d.tile == 0 and self.plan is None and not self.record_only
                and not torch.is_grad_enabled()):
            self._tuned[tune_key] = self._autotune(d, want_stats, pc.cout, mod, x.Cs)
            self._tune_alts[tune_key] = list(getattr(self, "_last_alts", []))
            self._tune_wide[tune_key] = list(getattr(self, "_last_wide", []))
            self._save_tune_cache()
            d.tile, d.splitk, d.prefetch = self._tuned[tune_key]
            pc = self._use_korder(d, mod, x.Cs, tile_korder(d.tile))
            d.bias = None if pc.bias is None else pc.bias.data_ptr()
            if want_stats:
                rows = lib.v2v_conv_stats_rows(C.byref(d))
                d.stats = self.scratch("stats", rows * pc.cout * 2).data_ptr()
        self._splitk_workspace(d)
        self._keep(pc.buf)
        check(lib.v2v_conv2d(C.byref(d), _stream()), "conv2d " + label)
        self.label(label)
        ntaps = pc.KH * pc.KW
        if len(self.conv_log) >= 100000:          # eager (training) use without a consumer: keep the log bounded
            del self.conv_log[:]
        self.conv_log.append(dict(label=label, N=N, H=H, W=W, OH=OH, OW=OW, cin=pc.cin, cout=pc.cout, tune_key=tune_key,
                                  KH=pc.KH, KW=pc.KW, stride=pc.stride, transposed=pc.transposed,
                                  flops=2.0 * N * (H * W if pc.transposed else OH * OW) * pc.cout * pc.cin * ntaps,
                                  tile=lib.v2v_conv_tile_config(C.byref(d)), splitk=max(int(d.splitk), 1),
                                  prefetch=int(d.prefetch), convs=2 if act_b is not None else 1))    # merged heads: two reference layers
        return out, rows, (N, OH, OW)

    # ---------------- paired launches (twin chains) ----------------
    def pair_eligible(self, xa, ma, xb, mb):
        """Can two 3x3 / stride 1 / pad 1 convolutions run as ONE v2v_conv2d_pair launch?  Identical geometry, channel
        stride a whole 128-byte chunk (the LDS-patch kernels' requirement), both behind a training-mode norm."""
        if self._training() or not isinstance(ma, nn.Conv2d) or not isinstance(mb, nn.Conv2d):
            return False
        bke = 64 if self.dtype == L.BF16 else 32
        same = (ma.in_channels == mb.in_channels and ma.out_channels == mb.out_channels and ma.kernel_size == mb.kernel_size
                and ma.stride == mb.stride and (ma.bias is None) == (mb.bias is None)
                and tuple(xa.t.shape) == tuple(xb.t.shape) and xa.Cs == xb.Cs and xa.C == xb.C)
        return bool(same and ma.kernel_size == (3, 3) and ma.stride == (1, 1) and xa.Cs % bke == 0)

    def _pair_desc(self, x, mod, pad_mode, pad, tile3, fin, label, fused=None):
        """Descriptor of one member of a paired launch: raw fp32 NHWC output + statistics (+ in-kernel finalize) on the
        CURRENT scratch sub-set.  fused = (act, act_param, res0, res1, y): the norm, activation and residual adds run in
        the conv kernel (V2V_OUT_NORM_ACT_NHWC) and `y` (an Act) receives the result; no raw tensor."""
        pc = self._use_korder1_pc(mod, x.Cs)
        N, H, W = x.N, x.H, x.W
        d = ConvDesc()
        d.in_ = x.t.data_ptr(); d.w = pc.buf.data_ptr(); d.w_korder = 1
        d.zero_page = self.zero_page().data_ptr()
        self._keep(self._zero_page)
        d.bias = None if pc.bias is None else pc.bias.data_ptr()
        d.N, d.H, d.W = N, H, W
        d.cin, d.cin_stride, d.cout = pc.cin, x.Cs, pc.cout
        d.KH, d.KW, d.stride, d.pad, d.pad_mode = 3, 3, 1, pad, pad_mode
        d.transposed = 0
        d.OH, d.OW = H, W
        d.dtype, d.out_mode, d.act = self.dtype, L.OUT_RAW_F32_NHWC, L.ACT_NONE
        d.act_param, d.out_scale = 0.0, 1.0
        d.ablate = self.ablate
        d.tile, d.splitk, d.prefetch = tile3
        if pc.cin != x.C:
            raise RuntimeError("conv %s: input has %d channels, layer expects %d" % (label, x.C, pc.cin))
        cs = (pc.cout + 3) // 4 * 4
        d.cout_stride = cs
        if fused is not None:
            act, act_param, res0, res1, y = fused
            raw = None
            d.out_mode, d.act, d.act_param = L.OUT_NORM_ACT_NHWC, act, act_param
            d.cout_stride = y.Cs
            d.out = y.t.data_ptr()
            d.res0 = None if res0 is None else res0.t.data_ptr()
            d.res1 = None if res1 is None else res1.t.data_ptr()
        else:
            raw = self.scratch("raw", N * H * W * cs)
            d.out = raw.data_ptr()
        d.stats = None
        rows = lib.v2v_conv_stats_rows(C.byref(d))
        if rows <= 0:
            check(rows or -1, "conv_stats_rows")
        if fused is not None:
            # tagged statistics granules (include/v2v_hip.h, "fused norm"): a zero-initialised buffer that only fused launches write
            d.stats = self.scratch("stats_tagged", rows * pc.cout * 4, zero=True).data_ptr()
        else:
            d.stats = self.scratch("stats", rows * pc.cout * 2).data_ptr()
        finalized = fin is not None and (fused is not None or N * H * W <= FUSE_FINALIZE_MAX_PIXELS)
        if finalized:
            norm, ss = fin
            gamma, beta, eps, mom, rm, rv = self._norm_params(norm, N)
            key = (self._lane, self._sset)
            fin_counter = self._fin_counters.get(key)
            if fin_counter is None:
                fin_counter = self._fin_counters[key] = torch.zeros(FIN_COUNTER_WORDS, dtype=torch.int32, device=self.device)
            d.fin_counter = fin_counter.data_ptr()
            d.fin_gamma = None if gamma is None else gamma.data_ptr()
            d.fin_beta = None if beta is None else beta.data_ptr()
            d.fin_scale_shift = ss.data_ptr()
            d.fin_running_mean = None if rm is None else rm.data_ptr()
            d.fin_running_var = None if rv is None else rv.data_ptr()
            d.fin_eps, d.fin_momentum, d.fin_count = eps, mom, N * H * W
            for t in (gamma, beta, ss, fin_counter):
                if t is not None:
                    self._keep(t)
        if pc.bias is not None:
            self._keep(pc.bias)
        if not self._splitk_workspace(d):
            raise RuntimeError("conv pair %s: split-K %d does not fit this layer" % (label, tile3[1]))
        self._keep(pc.buf)
        return d, raw, rows, finalized, pc

    def _use_korder1_pc(self, mod, cin_stride):
        pc = self.packed(mod, cin_stride, korder=1, refresh=False)
        if self.plan is None:
            pc.refresh()
        return pc

    def fused_norm_fits(self, tile3, N, H, W, cout, members=2):
        """V2V_OUT_NORM_ACT_NHWC (include/v2v_hip.h, "fused norm"): single-phase tiles, no split-K, every workgroup of the
        launch resident at once."""
        t, S = tile3[0], max(int(tile3[1]), 1)
        if not (self.fused_norm and self.fused_finalize and (80 <= t < 88 or 90 <= t < 94 or t in EXP_TILES) and S == 1 and cout % vec_of(self.dtype) == 0):
            return False
        th, tw, bn = PATCH_CFGS[t]
        if self._fused_norm_wgs is None:
            self._fused_norm_wgs = int(lib.v2v_conv_fused_norm_max_workgroups())
        return members * N * -(-H // th) * -(-W // tw) * -(-cout // bn) <= self._fused_norm_wgs

    # ---------------- bf16x3 ("x3") operands for the fp32 engine ----------------
    def _x3_ok(self, x, conv, pad_override=None, with_norm=True):
        tr = isinstance(conv, nn.ConvTranspose2d)
        if not (self.x3 and (isinstance(conv, nn.Conv2d) or (tr and tuple(conv.stride) == (2, 2))) and conv.groups == 1
                and conv.kernel_size in ((3, 3), (7, 7)) and x.C == conv.in_channels and x.C % 64 == 0 and x.Cs == x.C
                and self.fused_finalize and not self._training() and not self.record_only):
            return False
        if not with_norm:
            return True
        pad = conv.padding[0] if pad_override is None else pad_override
        st, k = conv.stride[0], conv.kernel_size[0]
        if tr:
            OH, OW = (x.H - 1) * st - 2 * pad + k + conv.output_padding[0], (x.W - 1) * st - 2 * pad + k + conv.output_padding[0]
        else:
            OH, OW = (x.H + 2 * pad - k) // st + 1, (x.W + 2 * pad - k) // st + 1
        return x.N * OH * OW <= FUSE_FINALIZE_MAX_PIXELS          # the statistics are finalized inside the conv launch

    def _x3_enter(self):
        sub = self._x3_engine
        if sub is None:
            sub = self._x3_engine = Engine(self.device, L.BF16, self.align_corners)
            sub.fused_norm = False                                # the norm runs in fp32 on the raw output (bn_apply of THIS engine)
        sub.plan, sub._lane, sub._sset = self.plan, self._lane, self._sset
        sub.autotune, sub.update_running_stats, sub.lanes_enabled = self.autotune, self.update_running_stats, self.lanes_enabled
        sub.fused_finalize = self.fused_finalize
        return sub

    def _x3_wrap(self, conv):
        w = self._x3_convs.get(id(conv))
        if w is None:
            w = self._x3_convs[id(conv)] = X3Conv(conv)
        return w

    def _x3_out(self, y):
        """The bf16x3 side output of a bn_apply that produces `y`, or None when `y` cannot feed an x3 convolution."""
        if not (self.x3 and y.C % 64 == 0 and y.Cs == y.C and not self._training() and not self.record_only):
            return None
        t = torch.empty((y.N, y.H, y.W, 3 * y.C), dtype=torch.bfloat16, device=self.device)
        self._keep(t)
        y.x3 = Act(t, 3 * y.C)
        return t

    def split_x3(self, x):
        """fp32 Act (C channels, C % 64 == 0, dense stride) -> bf16 Act of 3 C channels [hi | lo | hi]."""
        if x.x3 is not None:                                      # written by the producing bn_apply (v2v_bn_apply_x3)
            return x.x3
        out = torch.empty((x.N, x.H, x.W, 3 * x.C), dtype=torch.bfloat16, device=self.device)
        self._keep(out)
        check(lib.v2v_split_x3(_ptr(x.t), _ptr(out), x.N * x.H * x.W, x.C, x.Cs, 3 * x.C, _stream()), "split_x3")
        self.label("split_x3")
        return Act(out, 3 * x.C)

    def _x3_log(self, sub, n0):
        for c in sub.conv_log[n0:]:                               # algorithmic work in this engine's census (K was tripled)
            self.conv_log.append(dict(c, cin=c["cin"] // 3, flops=c["flops"] / 3.0, x3=True))
        del sub.conv_log[n0:]

    def conv_pair(self, xa, ma, xb, mb, pad_mode, pad, fins, labels, fuse=None):
        """Two convolutions of identical geometry as ONE launch (include/v2v_hip.h, v2v_conv2d_pair).  Member b works on
        scratch sub-set 1 of the current lane.  Returns ((raw, rows, finalized), (raw, rows, finalized)), shape.
        fuse = (act, act_param, adds_a, adds_b, ya, yb): when the selected tile allows it the norm / activation /
        residual adds run inside the launch and ya / yb receive the results (raw is then None)."""
        N, H, W = xa.N, xa.H, xa.W
        key = (-2, ma.in_channels, ma.out_channels, N, H, W, xa.Cs)
        if self.pair_override is not None:
            tile3 = (self.pair_override[0], self.pair_override[1], 0)
        elif key in self._tuned:
            tile3 = _cfg3(self._tuned[key])
        elif self.autotune and self.plan is None and not self.record_only and not torch.is_grad_enabled():
            tile3 = self._tuned[key] = self._autotune_pair(xa, ma, xb, mb, pad_mode, pad, fins, key, fuse=fuse)
            self._save_tune_cache()
        else:
            tile3 = (80 if W % 64 else 83, 1, 0)
        fa = fb = None
        if fuse is not None and fins[0] is not None and self.fused_norm_fits(tile3, N, H, W, ma.out_channels):
            act, act_param, adds_a, adds_b, ya, yb = fuse
            fa, fb = (act, act_param, adds_a[0], adds_a[1], ya), (act, act_param, adds_b[0], adds_b[1], yb)
        da, rawa, rowsa, fina, pca = self._pair_desc(xa, ma, pad_mode, pad, tile3, fins[0], labels[0], fused=fa)
        with self.scratch_set(1):
            db, rawb, rowsb, finb, pcb = self._pair_desc(xb, mb, pad_mode, pad, tile3, fins[1], labels[1], fused=fb)
        check(lib.v2v_conv2d_pair(C.byref(da), C.byref(db), _stream()), "conv2d_pair " + labels[0])
        self.label(labels[0] + " + " + labels[1])
        for lbl, pc in ((labels[0], pca), (labels[1], pcb)):
            self.conv_log.append(dict(label=lbl, N=N, H=H, W=W, OH=H, OW=W, cin=pc.cin, cout=pc.cout, tune_key=key,
                                      KH=3, KW=3, stride=1, transposed=False, pair=True, fused_norm=fa is not None,
                                      flops=2.0 * N * H * W * pc.cout * pc.cin * 9,
                                      tile=tile3[0], splitk=max(int(tile3[1]), 1), prefetch=0))
        return ((rawa, rowsa, fina), (rawb, rowsb, finb)), (N, H, W)

    def _autotune_pair(self, xa, ma, xb, mb, pad_mode, pad, fins, key, reps=7, fuse=None):
        """Measured choice of (tile, split-K) for a paired launch: every second-schedule ping-pong tile, unsplit and
        split 2 (cold weights: a 384 MB memset between launches, as _autotune)."""
        ncc = xa.Cs // (64 if self.dtype == L.BF16 else 32)
        if self._thrash is None:
            self._thrash = torch.empty(96 << 20, dtype=torch.float32, device=self.device)
        st = _stream()
        best, best_ms, alts = None, float("inf"), []
        for t in PAIR_TILES:
            th, tw, bn = PATCH_CFGS[t]
            tiles = xa.N * -(-xa.H // th) * -(-xa.W // tw) * -(-ma.out_channels // bn)
            for S in (1, 2):
                if S > 1 and (ncc < 2 * S or tiles * S * 2 > 1024):
                    continue
                fa = fb = None
                if fuse is not None and fins[0] is not None and self.fused_norm_fits((t, S, 0), xa.N, xa.H, xa.W, ma.out_channels):
                    act, act_param, adds_a, adds_b, ya, yb = fuse         # timed as it will run: norm / act / adds in the launch
                    fa, fb = (act, act_param, adds_a[0], adds_a[1], ya), (act, act_param, adds_b[0], adds_b[1], yb)
                try:
                    da = self._pair_desc(xa, ma, pad_mode, pad, (t, S, 0), fins[0], "tune", fused=fa)[0]
                    with self.scratch_set(1):
                        db = self._pair_desc(xb, mb, pad_mode, pad, (t, S, 0), fins[1], "tune", fused=fb)[0]
                except RuntimeError:
                    continue
                if lib.v2v_conv2d_pair(C.byref(da), C.byref(db), st) != 0:
                    continue
                e0 = [torch.cuda.Event(enable_timing=True) for _ in range(reps)]
                e1 = [torch.cuda.Event(enable_timing=True) for _ in range(reps)]
                for r in range(reps):
                    self._thrash.zero_()
                    e0[r].record()
                    lib.v2v_conv2d_pair(C.byref(da), C.byref(db), st)
                    e1[r].record()
                e1[-1].synchronize()
                ms = sorted(a.elapsed_time(b) for a, b in zip(e0, e1))[reps // 2]
                if fuse is not None and fa is None:
                    ms += 0.012                  # the bn_apply_pair launch the unfused variant still needs (~6 us + its gap)
                alts.append((ms, (t, S, 0)))
                if ms < best_ms:
                    best, best_ms = (t, S, 0), ms
        if best is None:
            raise RuntimeError("no paired-launch tile fits this layer")
        alts.sort()
        self._tune_alts[key] = [c for _, c in alts if c != best][:5]
        self._tune_wide[key] = []
        self.pair_tune_log = getattr(self, "pair_tune_log", {})
        self.pair_tune_log[key] = [(round(ms * 1e3, 1), c[0], c[1]) for ms, c in alts]
        return best

    def conv_group_pair(self, xa, conva, norma, xb, convb, normb, pad_mode, pad_override, act, act_param,
                        adds_a=(None, None), adds_b=(None, None), labels=("", "")):
        """Two [pad] conv + norm [+ act] [+ residuals] groups of identical geometry: ONE paired conv launch (statistics
        finalized in-kernel) + ONE paired normalise / activate / add launch.  Returns (Act a, Act b)."""
        cout = conva.out_channels
        ssa = self.scratch("scale_shift", 4 * cout)
        with self.scratch_set(1):
            ssb = self.scratch("scale_shift", 4 * cout)
        pad = conva.padding[0] if pad_override is None else pad_override
        fins = ((norma, ssa), (normb, ssb)) if self.fused_finalize else (None, None)
        N, OH, OW = xa.N, xa.H, xa.W
        ya, yb = self.empty_act(N, OH, OW, cout), self.empty_act(N, OH, OW, cout)
        if self._x3_ok(xa, conva, pad) and self._x3_ok(xb, convb, pad):
            sub = self._x3_enter()
            n0 = len(sub.conv_log)
            xa3 = self.split_x3(xa)
            with self.scratch_set(1):
                xb3 = self.split_x3(xb)
            (ra, rb), shp = sub.conv_pair(xa3, self._x3_wrap(conva), xb3, self._x3_wrap(convb), pad_mode, pad, fins, labels, fuse=None)
            self._x3_log(sub, n0)
        else:
            (ra, rb), shp = self.conv_pair(xa, conva, xb, convb, pad_mode, pad, fins, labels,
                                           fuse=(act, act_param, adds_a, adds_b, ya, yb))
        if ra[0] is None:                 # norm + activation + residuals ran inside the conv launch
            return ya, yb
        cs_raw = (cout + 3) // 4 * 4
        for (raw, rows, fin), norm, ss, k, lbl in ((ra, norma, ssa, 0, labels[0]), (rb, normb, ssb, 1, labels[1])):
            if not fin:                                       # large layers: parallel two-stage finalize per member
                with self.scratch_set(k):
                    gamma, beta, eps, mom, rm, rv = self._norm_params(norm, N)
                    st = self.scratch("stats", rows * cout * 2)
                    for t in (gamma, beta):
                        if t is not None:
                            self._keep(t)
                    groups = lib.v2v_bn_finalize_groups(rows)
                    ws = self.scratch("bn_ws", groups * cout * 2, torch.float64) if groups > 0 else None
                    check(lib.v2v_bn_finalize(_ptr(st), rows, cout, N * OH * OW, _ptr(gamma), _ptr(beta), eps,
                                              _ptr(ss), _ptr(rm), _ptr(rv), mom, _ptr(ws), _stream()), "bn_finalize " + lbl)
                    self.label(lbl + ".norm")
        a0, a1 = adds_a
        b0, b1 = adds_b
        xa3, xb3 = self._x3_out(ya), self._x3_out(yb)
        if xa3 is not None and xb3 is not None:
            check(lib.v2v_bn_apply_x3(_ptr(ra[0]), _ptr(ssa), _ptr(None if a0 is None else a0.t), _ptr(None if a1 is None else a1.t), _ptr(ya.t), _ptr(xa3),
                                      _ptr(rb[0]), _ptr(ssb), _ptr(None if b0 is None else b0.t), _ptr(None if b1 is None else b1.t), _ptr(yb.t), _ptr(xb3),
                                      cs_raw, N * OH * OW, cout, act, act_param, _stream()), "bn_apply_x3 (pair) " + labels[0])
        else:
            check(lib.v2v_bn_apply_pair(_ptr(ra[0]), _ptr(ssa), _ptr(None if a0 is None else a0.t), _ptr(None if a1 is None else a1.t), _ptr(ya.t),
                                        _ptr(rb[0]), _ptr(ssb), _ptr(None if b0 is None else b0.t), _ptr(None if b1 is None else b1.t), _ptr(yb.t),
                                        cs_raw, N * OH * OW, cout, ya.Cs, act, act_param, self.dtype, _stream()),
                  "bn_apply_pair " + labels[0])
        self.label(labels[0] + ".apply + " + labels[1] + ".apply")
        return ya, yb

    def run_resblocks_twin(self, blocks_a, xa, blocks_b, xb, name_a, name_b, first_index=0):
        """Two chains of ResnetBlocks of identical shape (models/networks.py:554-593) in lock step: every convolution pair
        is one launch that fills the chip (csrc/conv3x3_pp2_kernel.h), every normalise / residual pair one launch.
        Falls back to two independent chains when the pair is not eligible."""
        blocks_a, blocks_b = list(blocks_a), list(blocks_b)
        assert len(blocks_a) == len(blocks_b)
        for k, (ba, bb) in enumerate(zip(blocks_a, blocks_b)):
            ma, mb = list(ba.conv_block), list(bb.conv_block)
            def take(mods, j):
                pad_mode, pad_override = L.PAD_ZERO, None
                if isinstance(mods[j], nn.ReflectionPad2d):
                    pad_mode, pad_override = L.PAD_REFLECT, int(mods[j].padding[0]); j += 1
                return pad_mode, pad_override, mods[j], mods[j + 1], j + 2
            pm, po, c1a, n1a, ja = take(ma, 0)
            _, _, c1b, n1b, jb = take(mb, 0)
            na, nb = "%s.%d" % (name_a, first_index + k), "%s.%d" % (name_b, first_index + k)
            if not self.pair_eligible(xa, c1a, xb, c1b):
                xa = self.run_resblock(ba, xa, None, na)
                xb = self.run_resblock(bb, xb, None, nb)
                continue
            act, act_param = self._act_code(ma[ja]); ja += 1; jb += 1
            ha, hb = self.conv_group_pair(xa, c1a, n1a, xb, c1b, n1b, pm, po, act, act_param, labels=(na + ".c1", nb + ".c1"))
            pm, po, c2a, n2a, _ = take(ma, ja)
            _, _, c2b, n2b, _ = take(mb, jb)
            xa, xb = self.conv_group_pair(ha, c2a, n2a, hb, c2b, n2b, pm, po, L.ACT_NONE, 0.0,
                                          adds_a=(xa, None), adds_b=(xb, None), labels=(na + ".c2", nb + ".c2"))
        return xa, xb

    def tune_backward_data(self, d, dx_channels, conv=None, reflect=False):
        """Tile selection for a backward-data launch (autograd._conv_backward_data): same measured search as the
        forward convs (implicit-GEMM tiles x split-K; the operator is the forward kernel with role-swapped weights),
        keyed separately (leading -1) in the same tuning table.  Measured only while `autotune` is on (the first
        training steps of bench.py --mode train); later steps replay the selection."""
        key = (-1, d.cin, d.cout, d.KH, d.stride, d.transposed, d.N, d.H, d.W, d.cin_stride, d.cout_stride, d.pad)
        forced = self.bwd_tile_override.get((d.cin, d.cout, d.KH, d.stride, int(d.transposed)))
        if forced is not None:                       # tests: a given tile for the backward-data operator (dY channels, dX channels, k, stride, transposed)
            d.tile, d.splitk, d.prefetch = _cfg3(forced)
            if conv is not None:
                self._use_korder(d, conv, d.cin_stride, tile_korder(d.tile), role="bwd", reflect=reflect)
            if not self._splitk_workspace(d):
                raise RuntimeError("bwd_tile_override: split-K workspace")
            return
        if key not in self._tuned:
            if not (self.autotune and self.plan is None and not self.record_only):
                d.tile, d.splitk, d.prefetch = 0, 0, 0
                if conv is not None:
                    self._use_korder(d, conv, d.cin_stride, 0, role="bwd", reflect=reflect)
                self._splitk_workspace(d)
                return
            # conv given: the patch kernels are candidates too -- backward-data of a stride-2 Conv2d is a transposed stride-2 convolution
            # (conv3x3_t2_kernel), that of a ConvTranspose2d a stride-2 convolution (conv3x3_s2_kernel), on the role-swapped packings
            self._tuned[key] = self._autotune(d, False, dx_channels, mod=conv, cin_stride=d.cin_stride, role="bwd", reflect=reflect)
            self._save_tune_cache()
        d.tile, d.splitk, d.prefetch = _cfg3(self._tuned[key])
        if conv is not None:
            self._use_korder(d, conv, d.cin_stride, tile_korder(d.tile), role="bwd", reflect=reflect)
        if not self._splitk_workspace(d):
            d.tile, d.splitk, d.prefetch = 0, 0, 0
            if conv is not None:
                self._use_korder(d, conv, d.cin_stride, 0, role="bwd", reflect=reflect)
            self._splitk_workspace(d)

    def log_backward(self, kind, label, conv, cin, cout, N, pixels):
        """Algorithmic FLOP of a backward launch (same count as its forward conv), for bench.py's training roofline."""
        if len(self.conv_log) < 100000:
            self.conv_log.append(dict(label=kind + ":" + label, kind=kind, cin=cin, cout=cout, KH=conv.kernel_size[0],
                                      KW=conv.kernel_size[1], N=N, tile=-1, splitk=1, prefetch=0,
                                      flops=2.0 * N * pixels * cout * cin * conv.kernel_size[0] * conv.kernel_size[1]))

    def _save_tune_cache(self):
        if not self._tune_cache_path:
            return
        data = {}
        if os.path.exists(self._tune_cache_path):
            try:
                with open(self._tune_cache_path) as f:
                    data = json.load(f)
            except (OSError, ValueError):
                data = {}
        data[str(self.dtype)] = {",".join(str(int(x)) for x in k): list(v) for k, v in self._tuned.items()}
        with open(self._tune_cache_path, "w") as f:
            json.dump(data, f)

    def _use_korder1(self, d, mod, cin_stride):
        """Point the descriptor at the channel-chunk-major packing of `mod` (patch-kernel tile ids >= 32)."""
        pc = self.packed(mod, cin_stride, korder=1)
        d.w, d.w_korder = pc.buf.data_ptr(), 1
        return pc

    def bwd_patch_eligible(self, d, mod):
        """Backward-data of a 3x3 / stride 1 Conv2d on the single-phase 3x3 tiles (round 6): the operator IS a 3x3 convolution of the
        output gradient with the role-swapped, tap-flipped weights (PackedConv korder 4) and pad 2 - p -- behind a ReflectionPad2d
        (p = 0) a "full" convolution onto the padded grid, which reflect_pad_fold then folds.  The generic tiles ran these at
        62 us for the 1024 -> 1024 layers (the forward, same FLOP, takes 44 on tile 90: profiles/r06_v17_train_by_grid.txt)."""
        bke = 64 if self.dtype == L.BF16 else 32
        return (isinstance(mod, nn.Conv2d) and tuple(mod.kernel_size) == (3, 3) and tuple(mod.stride) == (1, 1) and mod.groups == 1
                and d.cin_stride % bke == 0 and d.out_mode == L.OUT_ACT_NHWC and os.environ.get("V2V_BWD_PATCH", "1") != "0")

    def bwd_c8_eligible(self, d, mod):
        """Backward-data of the 7x7 heads (ngf -> 3 behind ReflectionPad2d(3): models/networks.py:178-183) on conv7x7_c8_kernel
        (tile 61): the output gradient is ONE 16-byte vector per pixel, the operator a 7x7 convolution of it with the role-swapped,
        tap-flipped weights (PackedConv korder 5) and zero padding 6 - p.  The generic tiles walk it in 128-byte K chunks that are
        7/8 padding: 816 us per head at 2048x1024 (profiles/r06_v14_train_hires_by_grid.txt)."""
        vec = 8 if self.dtype == L.BF16 else 4
        return (isinstance(mod, nn.Conv2d) and tuple(mod.kernel_size) == (7, 7) and tuple(mod.stride) == (1, 1) and mod.groups == 1
                and d.cin_stride == vec and mod.in_channels <= 128 and mod.in_channels % vec == 0 and d.cout_stride % vec == 0
                and d.out_mode == L.OUT_ACT_NHWC and os.environ.get("V2V_BWD_C8", "1") != "0")

    def _use_korder(self, d, mod, cin_stride, korder, role="fwd", reflect=False):
        if role == "bwd" and d.tile == 61:
            if not self.bwd_c8_eligible(d, mod):
                raise ValueError("tile 61 as a backward-data operator: a 7x7 / stride 1 Conv2d whose output gradient is one 16-byte vector per pixel")
            pc = self.packed(mod, cin_stride, role="bwd", reflect=reflect, korder=5)
            d.transposed, d.pad = 0, pc.KH - 1 - pc.pad
            d.w, d.w_korder = pc.buf.data_ptr(), 0
            return pc
        if role == "bwd" and korder == 1 and 80 <= d.tile <= 93 and self.bwd_patch_eligible(d, mod):
            pc = self.packed(mod, cin_stride, role="bwd", reflect=reflect, korder=4)
            d.transposed, d.pad = 0, pc.KH - 1 - pc.pad
            d.w, d.w_korder = pc.buf.data_ptr(), 1
            return pc
        if d.tile in PERSISTENT_TILES and cin_stride == 32 and role == "fwd" and (self.pairx_eligible(d) or self.pairx_t_eligible(d)):
            # 64-byte pixels on the persistent single-chunk tiles: paired-x packing (PairedXConv).  Gated on the view's own
            # eligibility (round 6: an fp32 engine's 32-channel layers -- one 128-byte chunk, patch-eligible, so the
            # persistent tiles are among the candidates -- took this branch and PairedXConv raised for FlowNet2's 64 -> 32
            # layers at 512x256); everything else keeps the tile's own packing and the library refuses what it cannot run
            korder = 3
        pc = self.packed(mod, cin_stride, role=role, reflect=reflect, korder=korder)
        d.w, d.w_korder = pc.buf.data_ptr(), korder
        if role == "bwd":
            d.transposed, d.pad = int(pc.transposed), pc.pad      # (a previous candidate may have been the convolution form above)
        return pc

    def _use_korder0(self, d, mod, cin_stride):
        pc = self.packed(mod, cin_stride, korder=0)
        d.w, d.w_korder = pc.buf.data_ptr(), 0
        return pc

    def patch_eligible(self, d):
        bke = 64 if self.dtype == L.BF16 else 32
        return (not d.transposed and d.KH == 3 and d.KW == 3 and d.stride == 1 and d.pad == 1
                and d.cin_stride % bke == 0)

    def pairx_eligible(self, d):
        """Persistent single-chunk tiles 140 / 141 (/ 143) on a layer with 64-byte pixels (<= 32 -> 32 channels, bf16, raw fp32 output): the
        paired-x view (PairedXConv) -- pairs of pixels as one 128-byte pixel of a 64 -> 64 layer."""
        return (self.dtype == L.BF16 and not d.transposed and d.KH == 3 and d.KW == 3 and d.stride == 1 and d.pad == 1
                and d.cin_stride == 32 and d.cout == 32 and d.out_mode in (L.OUT_RAW_F32_NHWC, L.OUT_RAW_ACT_NHWC) and d.W % 2 == 0 and (d.W // 2) % 32 == 0
                and d.H % 8 == 0 and os.environ.get("V2V_PAIRX", "1") != "0")

    def pairx_t_eligible(self, d):
        """Persistent transposed tile 114 on a layer with 64-byte pixels (<= 32 -> 16 channels): the paired-x view (PairedXConvT)."""
        return (self.dtype == L.BF16 and bool(d.transposed) and d.KH == 3 and d.KW == 3 and d.stride == 2 and d.pad == 1
                and d.cin_stride == 32 and d.cout == 16 and d.out_mode in (L.OUT_RAW_F32_NHWC, L.OUT_RAW_ACT_NHWC) and d.W % 2 == 0 and (d.W // 2) % 32 == 0
                and d.H % 8 == 0 and d.OH == 2 * d.H and d.OW == 2 * d.W and os.environ.get("V2V_PAIRX", "1") != "0")

    def s7_eligible(self, d):
        """7x7-window tiles 120 / 121: dense bf16 7x7 / stride 1 / pad 3 Conv2d whose channel stride is a whole number of 128-byte chunks
        (the stems on the pooled label encodings, edge2face's 45 -> 128 stem).  Round 5: validated on three boxes (parity test, 1.1-1.7x on
        the 64 / 128-channel stems: profiles/r05_v1_stem7_bench.txt) -- ON by default, V2V_S7_PATCH=0 takes them out of the search."""
        return (self.dtype == L.BF16 and not d.transposed and d.KH == 7 and d.KW == 7 and d.stride == 1 and d.pad == 3
                and d.cin_stride % 64 == 0 and d.out_mode != L.OUT_NORM_ACT_NHWC and os.environ.get("V2V_S7_PATCH", "1") != "0")

    def s2_eligible(self, d):
        """conv3x3_s2_kernel: 3x3 / stride 2 / zero pad 1 Conv2d whose channel stride is a whole 128-byte chunk."""
        bke = 64 if self.dtype == L.BF16 else 32
        return (not d.transposed and d.KH == 3 and d.KW == 3 and d.stride == 2 and d.pad == 1 and d.pad_mode == L.PAD_ZERO
                and d.cin_stride % bke == 0 and d.out_mode != L.OUT_NORM_ACT_NHWC and os.environ.get("V2V_S2_PATCH", "1") != "0")

    def t2_eligible(self, d):
        """conv3x3_t2_kernel: ConvTranspose2d(3x3, stride 2, padding 1) whose channel stride is a whole 128-byte chunk."""
        bke = 64 if self.dtype == L.BF16 else 32
        return (bool(d.transposed) and d.KH == 3 and d.KW == 3 and d.stride == 2 and d.pad == 1
                and d.cin_stride % bke == 0 and d.out_mode != L.OUT_NORM_ACT_NHWC and os.environ.get("V2V_T2_PATCH", "1") != "0")

    def _splitk_workspace(self, d):
        """Attach the split-K slab scratch and ticket words to a descriptor (no-op for splitk <= 1)."""
        if d.splitk <= 1:
            d.splitk = 0
            d.slabs = None
            d.sk_counter = None
            return True
        tickets = C.c_int32(0)
        nbytes = lib.v2v_conv_splitk_workspace(C.byref(d), C.byref(tickets))
        if nbytes <= 0:
            return False
        skc = self._sk_counters.get((self._lane, self._sset))
        if skc is None or skc.numel() < tickets.value:
            if self.plan is not None and not self.record_only and skc is not None:      # (a first buffer of this lane is safe: nothing recorded points at one yet)
                raise RuntimeError("split-K ticket buffer must not grow while a plan is recording")
            skc = self._sk_counters[(self._lane, self._sset)] = torch.zeros(max(4096, tickets.value), dtype=torch.int32, device=self.device)
        d.slabs = self.scratch("slabs", (nbytes + 3) // 4).data_ptr()
        d.sk_counter = skc.data_ptr()
        self._keep(skc)
        return True

    def _autotune(self, d, want_stats, cout, mod=None, cin_stride=0, reps=5, role="fwd", reflect=False):
        """Time every (tile, split-K, weight-prefetch) configuration that fits this launch; returns the fastest
        triple.  Runs once per conv shape while a frame plan is being built, never inside a timed region.  Each
        timed launch is preceded by a 384 MB memset: at batch 1 a frame streams ~0.8 GB of weights, so every layer
        meets its weights COLD in L2 / Infinity Cache -- back-to-back warm launches favour shallow LDS-DMA rings
        and no prefetch, which stall on real frames."""
        M = d.N * (d.H * d.W if d.transposed else d.OH * d.OW)
        ncls = 4 if (d.transposed and d.stride == 2) else 1
        nk = (d.KH * d.KW * d.cin_stride * (2 if self.dtype == L.BF16 else 4)) // 128 // ncls
        cands = []
        for t, (bm, bn, helper) in sorted(TILE_CFGS.items()):
            if t == 4 and cout > 32:
                continue
            tiles = -(-M // (bm * ncls)) * -(-cout // bn) * ncls
            for S in (1, 2, 3, 4, 6, 8):
                if S > 1 and (tiles * S > 1024 or nk // S < 4):
                    continue
                if S == 1 and t >= 18 and tiles < 96:
                    continue            # large tiles that cannot fill the chip without split-K
                cands.append((t, S, 0))      # (prefetch-helper variants never won a sweep: not searched)
        bke_ = 32 if self.dtype == L.BF16 else 16         # whole 64-byte half chunks (conv7x7_head_kernel, HC = 1)
        if (not d.transposed and d.KH == 7 and d.KW == 7 and d.stride == 1 and d.pad == 3 and cout <= 32
                and d.cin_stride % bke_ == 0 and not d.fin_counter
                and (d.out_mode == L.OUT_F32_NCHW or d.out_mode == L.OUT_RAW_F32_NHWC)):
            cands.append((60, 1, 0))          # conv7x7_head_kernel (LDS patch + 16-wide MFMA)
            if (self.rowsum_heads and self.dtype == L.BF16 and cout <= 4 and d.out_mode == L.OUT_F32_NCHW and d.cin_stride % 32 == 0
                    and not d.stats):
                cands.append((62, 1, 0))      # conv7x7_rowsum_kernel (row GEMM + shifted sum)
        if (not d.transposed and d.KH == 7 and d.KW == 7 and d.stride == 1 and d.pad == 3 and cout <= 128
                and d.cin_stride * (2 if self.dtype == L.BF16 else 4) == 16 and not d.fin_counter
                and (d.out_mode == L.OUT_F32_NCHW or d.out_mode == L.OUT_RAW_F32_NHWC)):
            cands.append((61, 1, 0))          # conv7x7_c8_kernel: 16-byte pixels (the 6-channel previous-frame stems), four taps per MFMA step
        if mod is not None and role == "fwd" and self.patch_eligible(d):
            ncc = d.cin_stride // (64 if self.dtype == L.BF16 else 32)
            for t, (th, tw, bn) in sorted(PATCH_CFGS.items()):
                if t in EXP_TILES and os.environ.get("V2V_EXP_TILES", "0") != "1":
                    continue
                if tw == 64 and d.OW % 64 != 0 and d.OW > 32:
                    pass                      # ragged tiles are legal, just wasteful; let the timing decide
                tiles = d.N * -(-d.OH // th) * -(-d.OW // tw) * -(-cout // bn)
                for S in (1, 2, 3, 4, 8):
                    if S > 1 and (tiles * S > 1024 or ncc < S):
                        continue
                    if S == 1 and tiles < 64:
                        continue
                    cands.append((t, S, 0))
        if mod is not None and role == "bwd" and self.bwd_c8_eligible(d, mod):
            cands.append((61, 1, 0))          # conv7x7_c8_kernel on the tap-flipped role-swapped weights (full convolution of the head's output gradient)
        if mod is not None and role == "bwd" and self.bwd_patch_eligible(d, mod):
            for t in (80, 81, 82, 83, 84, 85, 86, 87, 90, 91, 92, 93):
                th, tw, bn = PATCH_CFGS[t]
                tiles = d.N * -(-d.OH // th) * -(-d.OW // tw) * -(-cout // bn)
                ncc = d.cin_stride // (64 if self.dtype == L.BF16 else 32)
                for S in (1, 2):
                    if (S > 1 and (tiles * S > 1024 or ncc < S)) or (S == 1 and tiles < 64):
                        continue
                    cands.append((t, S, 0))
        if mod is not None and role == "fwd" and self.pairx_eligible(d) and d.N * (d.H // 8) * (d.W // 64) >= 64:
            for t in ONE_TILES:
                if t not in EXP_TILES or os.environ.get("V2V_EXP_TILES", "0") == "1":
                    cands.append((t, 1, 0))
        if mod is not None and role == "fwd" and self.pairx_t_eligible(d) and d.N * (d.H // 8) * (d.W // 64) >= 48:
            cands.append((114, 1, 0))
        if mod is not None and role == "fwd" and self.s7_eligible(d):
            for t, (th, tw, bn) in sorted(S7_CFGS.items()):
                tiles = d.N * -(-d.OH // th) * -(-d.OW // tw) * -(-cout // bn)
                if tiles >= 64 and (bn <= 64 or cout > 64):
                    cands.append((t, 1, 0))
        if mod is not None and self.s2_eligible(d):
            for t, (th, tw, bn) in sorted(S2_CFGS.items()):
                tiles = d.N * -(-d.OH // th) * -(-d.OW // tw) * -(-cout // bn)
                if tiles >= 64:
                    cands.append((t, 1, 0))
        if mod is not None and self.t2_eligible(d):
            for t, (th, tw, bn) in sorted(T2_CFGS.items()):
                tiles = d.N * -(-((d.OH + 1) // 2) // th) * -(-((d.OW + 1) // 2) // tw) * -(-cout // bn)
                if tiles >= 48:
                    cands.append((t, 1, 0))
        st = _stream()
        if self._thrash is None:
            self._thrash = torch.empty(96 << 20, dtype=torch.float32, device=self.device)
        def time_cfg(t, S, pf, reps):
            d.tile, d.splitk, d.prefetch = t, S, pf
            if mod is not None:
                self._use_korder(d, mod, cin_stride, tile_korder(t), role=role, reflect=reflect)
            if want_stats:
                rows = lib.v2v_conv_stats_rows(C.byref(d))
                if rows <= 0:
                    return None
                d.stats = self.scratch("stats", rows * cout * 2).data_ptr()
            if not self._splitk_workspace(d):
                return None
            if lib.v2v_conv2d(C.byref(d), st) != 0:
                return None
            e0 = [torch.cuda.Event(enable_timing=True) for _ in range(reps)]
            e1 = [torch.cuda.Event(enable_timing=True) for _ in range(reps)]
            for r in range(reps):
                self._thrash.zero_()
                e0[r].record()
                lib.v2v_conv2d(C.byref(d), st)
                e1[r].record()
            e1[-1].synchronize()
            return sorted(a.elapsed_time(b) for a, b in zip(e0, e1))[reps // 2]      # median

        timed = []
        for t, S, pf in cands:
            ms = time_cfg(t, S, pf, 3)
            if ms is not None:
                timed.append((ms, (t, S, pf)))
        if os.environ.get("V2V_TUNE_DEBUG"):
            print("autotune %s: %d candidates, %d ran; fastest %s; last error: %s" % ((d.cin, d.cout, d.KH, d.stride, d.transposed, d.cout_stride), len(cands), len(timed),
                  sorted(timed)[:3], lib.v2v_last_error().decode()[:200]), file=sys.stderr)
        if not timed:
            return (0, 1, 0)
        # second pass over the front-runners with more repetitions: single medians of 3 are noisy enough to flip
        # the choice between runs (181 vs 194 fps observed)
        timed.sort()
        best, best_ms = timed[0][1], float("inf")
        for _, cfg in timed[:6]:
            ms = time_cfg(cfg[0], cfg[1], cfg[2], 11)
            if ms is not None and ms < best_ms:
                best, best_ms = cfg, ms
        # runners-up for the whole-frame search of the frame plan (models/vid2vid_model_G._FramePlan._frame_tune): the
        # fastest few in isolation plus the fastest unsplit ones (split-K fills an idle chip; beside concurrent lanes it
        # only adds slab traffic)
        unsplit = [cfg for _, cfg in timed if cfg[1] <= 1]
        alts = ([cfg for _, cfg in timed[:3]] + unsplit[:3]
                + [cfg for _, cfg in timed if cfg[0] == best[0] and cfg[1] <= 2]          # the winner's tile, less split
                + [cfg for cfg in unsplit if 50 <= cfg[0] < 60 or cfg[0] >= 70][:1])         # the best unsplit ping-pong tile
        self._last_alts = [c for i, c in enumerate(alts) if c != best and c not in alts[:i]][:7]
        # for the heaviest shapes of a frame the whole-frame search also walks every lightly split configuration that was
        # not hopeless in isolation
        pp_ = lambda t: 50 <= t < 60 or t >= 70
        self._last_wide = ([cfg for ms, cfg in timed if pp_(cfg[0]) and cfg[1] <= 2 and cfg != best]      # every ping-pong tile
                           + [cfg for ms, cfg in timed if not pp_(cfg[0]) and cfg[1] <= 2 and ms <= 1.7 * timed[0][0]
                              and cfg != best][:10])
        return best

    def _norm_params(self, norm, N):
        """(gamma, beta, eps, momentum, running_mean, running_var) of a training-mode norm layer
        (get_norm_layer, models/networks.py:23-30; .eval() is never called in the reference)."""
        if isinstance(norm, nn.BatchNorm2d):
            gamma = norm.weight.detach() if norm.affine else None
            beta = norm.bias.detach() if norm.affine else None
            eps, mom = norm.eps, (norm.momentum if norm.momentum is not None else 0.1)
            rm = norm.running_mean if (self.update_running_stats and norm.track_running_stats) else None
            rv = norm.running_var if (self.update_running_stats and norm.track_running_stats) else None
            if rm is not None and self.plan is None:
                norm._v2v_batches = getattr(norm, "_v2v_batches", 0) + 1     # -> num_batches_tracked at save time
            return gamma, beta, eps, mom, rm, rv
        if isinstance(norm, nn.InstanceNorm2d):
            if N != 1:
                raise NotImplementedError("InstanceNorm2d path supports batch 1 (per-sample statistics)")
            gamma = norm.weight.detach() if norm.affine else None
            beta = norm.bias.detach() if norm.affine else None
            return gamma, beta, norm.eps, 0.1, None, None
        raise NotImplementedError("norm layer %r" % type(norm))

    def norm_apply(self, raw, rows, shape, cout, norm, act, act_param, add0=None, add1=None, label="", ss=None,
                   finalized=False):
        """[bn_finalize +] bn_apply on the shared raw/statistics scratch (ss: caller-owned [4][C] statistics
        buffer, kept for the backward pass on the training path; finalized: the conv kernel already wrote it)."""
        N, OH, OW = shape
        raw_bf16 = raw.dtype == torch.bfloat16                 # Engine.conv(raw_act_ok=True): V2V_OUT_RAW_ACT_NHWC
        cs_raw = (cout + 7) // 8 * 8 if raw_bf16 else (cout + 3) // 4 * 4
        if ss is None:
            ss = self.scratch("scale_shift", 4 * cout)
        if not finalized:
            gamma, beta, eps, mom, rm, rv = self._norm_params(norm, N)
            st = self.scratch("stats", rows * cout * 2)
            for t in (gamma, beta):
                if t is not None:
                    self._keep(t)
            groups = lib.v2v_bn_finalize_groups(rows)
            ws = self.scratch("bn_ws", groups * cout * 2, torch.float64) if groups > 0 else None
            check(lib.v2v_bn_finalize(_ptr(st), rows, cout, N * OH * OW, _ptr(gamma), _ptr(beta), eps,
                                      _ptr(ss), _ptr(rm), _ptr(rv), mom, _ptr(ws), _stream()), "bn_finalize " + label)
            self.label(label + ".norm")
        y = self.empty_act(N, OH, OW, cout)
        y3 = self._x3_out(y)
        if y3 is not None:
            check(lib.v2v_bn_apply_x3(_ptr(raw), _ptr(ss), _ptr(None if add0 is None else add0.t), _ptr(None if add1 is None else add1.t),
                                      _ptr(y.t), _ptr(y3), None, None, None, None, None, None,
                                      cs_raw, N * OH * OW, cout, act, act_param, _stream()), "bn_apply_x3 " + label)
        elif raw_bf16:
            check(lib.v2v_bn_apply_raw(_ptr(raw), L.BF16, cs_raw, _ptr(ss),
                                       _ptr(None if add0 is None else add0.t), _ptr(None if add1 is None else add1.t),
                                       _ptr(y.t), N * OH * OW, cout, y.Cs, act, act_param, self.dtype, _stream()),
                  "bn_apply_raw " + label)
        else:
            check(lib.v2v_bn_apply(_ptr(raw), cs_raw, _ptr(ss),
                                   _ptr(None if add0 is None else add0.t), _ptr(None if add1 is None else add1.t),
                                   _ptr(y.t), N * OH * OW, cout, y.Cs, act, act_param, self.dtype, _stream()),
                  "bn_apply " + label)
        self.label(label + ".apply")
        return y

    def conv_group(self, x, conv, pad_mode, pad_override, norm, act, act_param, add0=None, add1=None,
                   head_nchw=False, out_scale=1.0, label=""):
        """One fused [pad] conv [norm] [act] [+ residuals] group.  Returns an Act, or the planar fp32 NCHW
        tensor when head_nchw.  With autograd recording on (training) it is a v2v custom op
        (autograd.ConvFn: same launches, tensors saved for the HIP backward kernels)."""
        if self.plan is None and torch.is_grad_enabled():
            from . import autograd as AG
            return AG.conv_group(self, x, conv, pad_mode, pad_override, norm, act, act_param, add0, add1,
                                 head_nchw, out_scale, label)
        if norm is not None and self.onehot_eligible(x, conv, pad_mode, pad_override):
            ss = self.scratch("scale_shift", 4 * conv.out_channels)
            fin = (norm, ss) if self.fused_finalize else None
            raw, rows, shp = self.onehot_conv(x, conv, label=label, fin=fin)
            return self.norm_apply(raw, rows, shp, conv.out_channels, norm, act, act_param, add0=add0, add1=add1, label=label,
                                   ss=ss, finalized=fin is not None)
        if norm is not None:
            ss = self.scratch("scale_shift", 4 * conv.out_channels)
            if self._x3_ok(x, conv, pad_override):
                sub = self._x3_enter()
                n0 = len(sub.conv_log)
                raw, rows, shp = sub.conv(self.split_x3(x), self._x3_wrap(conv), pad_mode, pad_override, L.OUT_RAW_F32_NHWC,
                                          want_stats=True, label=label, fin=(norm, ss))
                self.last_finalized = sub.last_finalized
                self._x3_log(sub, n0)
            else:
                raw, rows, shp = self.conv(x, conv, pad_mode, pad_override, L.OUT_RAW_F32_NHWC, want_stats=True, label=label,
                                           fin=(norm, ss) if self.fused_finalize else None, raw_act_ok=True)
            return self.norm_apply(raw, rows, shp, conv.out_channels, norm, act, act_param, add0=add0, add1=add1,
                                   label=label, ss=ss, finalized=self.fused_finalize and self.last_finalized)
        if add0 is not None or add1 is not None:
            raise NotImplementedError("residual adds need a norm layer in the group")
        bke = 32 if self.dtype == L.BF16 else 16              # the 7x7 head kernel reads whole 64-byte half chunks
        if (head_nchw and isinstance(conv, nn.Conv2d) and conv.kernel_size == (7, 7) and conv.out_channels <= 32
                and x.Cs % bke != 0 and x.H * x.W >= 65536):
            x = self.widen(x, (x.Cs + bke - 1) // bke * bke)      # e.g. the 16-channel foreground tower of scale 2: 32-byte rows
        if head_nchw and self._x3_ok(x, conv, pad_override, with_norm=False):     # API-facing 7x7 heads: planar fp32 straight from the bf16x3 product
            sub = self._x3_enter()
            n0 = len(sub.conv_log)
            out, _, _ = sub.conv(self.split_x3(x), self._x3_wrap(conv), pad_mode, pad_override, L.OUT_F32_NCHW, act, act_param, out_scale, label=label)
            self._x3_log(sub, n0)
            return out
        out, _, _ = self.conv(x, conv, pad_mode, pad_override, L.OUT_F32_NCHW if head_nchw else L.OUT_ACT_NHWC,
                              act, act_param, out_scale if head_nchw else 1.0, label=label)
        return out

    def head_pair(self, x, seq_a, scale_a, seq_b, scale_b, label=""):
        """Two API-facing heads on the same input ([ReflectionPad2d(3), Conv2d(C, n, 7) [, act]] each: model_final_flow and
        model_final_w, models/networks.py:181-183) as ONE launch writing one planar fp32 tensor; returns the two channel
        ranges as views.  None when the pair does not qualify (the caller then emits them one by one)."""
        def parse(seq):
            mods = list(seq)
            if not (len(mods) in (2, 3) and isinstance(mods[0], nn.ReflectionPad2d) and isinstance(mods[1], nn.Conv2d)):
                return None
            act = (L.ACT_NONE, 0.0) if len(mods) == 2 else self._act_code(mods[2])
            return None if act is None else (int(mods[0].padding[0]), mods[1], act)
        pa, pb = parse(seq_a), parse(seq_b)
        if (not self.merge_heads or self.x3 or pa is None or pb is None or self._training() or pa[0] != pb[0]
                or pa[1].kernel_size != (7, 7) or pa[1].out_channels + pb[1].out_channels > 16):
            return None
        key = (id(pa[1]), id(pb[1]))
        merged = self._merged_heads.get(key)
        if merged is None:
            try:
                merged = self._merged_heads[key] = MergedConv(pa[1], pb[1])
            except ValueError:
                return None
        bke = 32 if self.dtype == L.BF16 else 16
        if x.Cs % bke != 0:                       # the 7x7 head kernel reads whole 64-byte half chunks
            if x.H * x.W < 65536:
                return None
            x = self.widen(x, (x.Cs + bke - 1) // bke * bke)
        ca = pa[1].out_channels
        out, _, _ = self.conv(x, merged, L.PAD_REFLECT, pa[0], L.OUT_F32_NCHW, pa[2][0], pa[2][1], scale_a, label=label,
                              act_b=(ca, pb[2][0], pb[2][1], scale_b))
        return out[:, :ca], out[:, ca:]

    # ---------------- nn.Sequential lowering ----------------
    @staticmethod
    def _act_code(m):
        if isinstance(m, nn.ReLU): return L.ACT_RELU, 0.0
        if isinstance(m, nn.LeakyReLU): return L.ACT_LEAKY, float(m.negative_slope)
        if isinstance(m, nn.Tanh): return L.ACT_TANH, 0.0
        if isinstance(m, nn.Sigmoid): return L.ACT_SIGMOID, 0.0
        return None

    def run_sequential(self, seq, x, extra_add=None, head_nchw=False, out_scale=1.0, name="", collect=None, first_index=0):
        """Run an nn.Sequential (or list of modules) on Act `x`.

        extra_add: Act added to the output of the LAST stage (tower sums, coarse features).
        head_nchw: the last conv writes planar fp32 NCHW (API-facing heads).
        collect:   optional list receiving the output of every top-level module group
                   (MultiscaleDiscriminator.getIntermFeat).
        first_index: index of seq[0] inside the nn.Sequential it was sliced from (labels keep the reference's numbering).
        """
        mods = list(seq)
        i, n = 0, len(mods)
        while i < n:
            m = mods[i]
            last_group = False
            if isinstance(m, (nn.ReflectionPad2d, nn.Conv2d, nn.ConvTranspose2d)):
                pad_mode, pad_override = L.PAD_ZERO, None
                if isinstance(m, nn.ReflectionPad2d):
                    pad_mode, pad_override = L.PAD_REFLECT, int(m.padding[0])
                    i += 1
                    m = mods[i]
                    if not isinstance(m, nn.Conv2d) or m.padding[0] != 0:
                        raise NotImplementedError("ReflectionPad2d must be followed by an unpadded Conv2d")
                conv = m
                i += 1
                norm = None
                if i < n and isinstance(mods[i], (nn.BatchNorm2d, nn.InstanceNorm2d)):
                    norm = mods[i]; i += 1
                act, act_param = L.ACT_NONE, 0.0
                if i < n and self._act_code(mods[i]) is not None:
                    act, act_param = self._act_code(mods[i]); i += 1
                last_group = i >= n
                lbl = "%s.%d" % (name, first_index + i)
                if norm is not None:
                    x = self.conv_group(x, conv, pad_mode, pad_override, norm, act, act_param,
                                        add0=extra_add if last_group else None, label=lbl)
                    if last_group:
                        extra_add = None
                elif head_nchw and last_group:
                    return self.conv_group(x, conv, pad_mode, pad_override, None, act, act_param,
                                           head_nchw=True, out_scale=out_scale, label=lbl)
                else:
                    x = self.conv_group(x, conv, pad_mode, pad_override, None, act, act_param, label=lbl)
            elif hasattr(m, "conv_block"):      # ResnetBlock (models/networks.py:554-593)
                i += 1
                last_group = i >= n
                x = self.run_resblock(m, x, extra_add if last_group else None, "%s.%d" % (name, first_index + i - 1))
                if last_group:
                    extra_add = None
            elif isinstance(m, nn.Sequential):
                i += 1
                x = self.run_sequential(m, x, name="%s.%d" % (name, first_index + i - 1))
            elif isinstance(m, nn.Dropout):
                raise NotImplementedError("dropout is never enabled on the vid2vid path")
            else:
                raise NotImplementedError("module %r in %s" % (type(m), name))
            if collect is not None:
                collect.append(x)
        if extra_add is not None:
            x = self.add(x, extra_add)
        return x

    def run_resblock(self, blk, x, extra_add, name):
        mods = list(blk.conv_block)
        # [pad, conv, norm, act, pad, conv, norm]   (reflect) -- zero padding variant has no pad modules
        def take(j):
            pad_mode, pad_override = L.PAD_ZERO, None
            if isinstance(mods[j], nn.ReflectionPad2d):
                pad_mode, pad_override = L.PAD_REFLECT, int(mods[j].padding[0]); j += 1
            elif isinstance(mods[j], nn.ReplicationPad2d):
                raise NotImplementedError("replicate padding is not used by vid2vid")
            conv = mods[j]; norm = mods[j + 1]
            return pad_mode, pad_override, conv, norm, j + 2
        pm, po, conv1, norm1, j = take(0)
        act, act_param = self._act_code(mods[j]); j += 1
        h = self.conv_group(x, conv1, pm, po, norm1, act, act_param, label=name + ".c1")
        pm, po, conv2, norm2, j = take(j)
        return self.conv_group(h, conv2, pm, po, norm2, L.ACT_NONE, 0.0, add0=x, add1=extra_add, label=name + ".c2")

    # ---------------- other ops ----------------
    def _training(self):
        return self.plan is None and torch.is_grad_enabled()      # (dry-run engines build the autograd graph too)

    def add(self, a, b):
        if a.t.shape != b.t.shape:
            raise RuntimeError("add: shape mismatch %s vs %s" % (tuple(a.t.shape), tuple(b.t.shape)))
        if self._training() and (a.t.requires_grad or b.t.requires_grad):
            from . import autograd as AG
            return Act(AG.AddFn.apply(self, a.t, b.t), a.C)
        y = self.empty_act(a.N, a.H, a.W, a.C)
        check(lib.v2v_add_nhwc(_ptr(a.t), _ptr(b.t), _ptr(y.t), a.t.numel(), self.dtype, _stream()), "add")
        self.label("add_nhwc")
        return y

    def _fg_labels(self, fg_labels):
        """The foreground label ids on the device, built once per id list: torch.tensor(list, device=...) is a synchronous copy
        from pageable memory -- not permitted inside a stream capture (graphed.ChunkGraphs), and a host round trip per frame."""
        key = tuple(int(v) for v in fg_labels)
        cache = self.__dict__.setdefault("_fg_label_cache", {})
        t = cache.get(key)
        if t is None:
            t = cache[key] = torch.tensor(list(key), dtype=torch.int32, device=self.device)
        return t

    def fg_mask(self, x, base_ch, fg_labels):
        mask = self.empty_f32(x.N, 1, x.H, x.W)
        fg = self._fg_labels(fg_labels)
        self._keep(fg)
        check(lib.v2v_fg_mask_nhwc(_ptr(x.t), _ptr(mask), x.N * x.H * x.W, x.Cs, base_ch, _ptr(fg), fg.numel(),
                                   self.dtype, _stream()), "fg_mask")
        self.label("fg_mask_nhwc")
        return mask

    def memcpy(self, dst, src, nbytes):
        check(lib.v2v_memcpy_d2d(_ptr(dst), _ptr(src), nbytes, _stream()), "memcpy_d2d")
        self.label("memcpy_d2d")

    def encode_labels_pooled(self, labels, inst, T, H, W, label_nc, fg_labels, want_mask, chunk_stride=False, source=None):
        """(lazy full-resolution Act, pooled Act, full-resolution mask): the encoded label maps one pyramid level down, straight
        from the maps (v2v_encode_labels_pooled).  The full-resolution Act carries only its LabelSource -- its tensor is allocated
        (shape / stride bookkeeping) but NEVER WRITTEN: legal only when every consumer reads the maps (gather-sum stems)."""
        per = label_nc + (1 if inst is not None else 0)
        bke = 64 if self.dtype == L.BF16 else 32
        cs = pad_channels(T * per, self.dtype)
        if chunk_stride:
            cs = (T * per + bke - 1) // bke * bke
        lazy = torch.empty((1, H, W, cs), dtype=self.tdtype, device=self.device)
        OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        pooled = torch.empty((1, OH, OW, cs), dtype=self.tdtype, device=self.device)
        self._keep(pooled)
        mask = self.empty_f32(1, 1, H, W) if want_mask else None
        fg = None
        if want_mask:
            fg = self._fg_labels(fg_labels)
            self._keep(fg)
        u8 = labels.dtype == torch.uint8
        if u8 and inst is not None and inst.dtype != torch.int32:
            raise TypeError("uint8 label maps go with int32 instance maps")
        if not u8 and (labels.dtype != torch.float32 or (inst is not None and inst.dtype != torch.float32)):
            raise TypeError("label / instance maps must be fp32-encoded integers, or uint8 + int32")
        codes = getattr(source, "codes", None) if source is not None else None
        if codes is not None and os.environ.get("V2V_POOLED_FROM_CODES", "1") != "0":
            # the frame plan has the 1-byte label | edge codes already (label_codes, for the gather-sum stems): nine independent byte
            # loads per frame instead of a chain of label / instance-map loads (394 -> see profiles/r05_*): same bits
            self._keep(codes)
            check(lib.v2v_encode_labels_pooled(_ptr(codes), _ptr(inst), _ptr(pooled), _ptr(mask), T, H, W, label_nc, cs, _ptr(fg),
                                               0 if fg is None else fg.numel(), self.dtype, 2, _stream()), "encode_labels_pooled (codes)")
        else:
            check(lib.v2v_encode_labels_pooled(_ptr(labels), _ptr(inst), _ptr(pooled), _ptr(mask), T, H, W, label_nc, cs, _ptr(fg),
                                               0 if fg is None else fg.numel(), self.dtype, int(u8), _stream()), "encode_labels_pooled")
        self.label("encode_labels_pooled")
        x0 = Act(lazy, T * per)
        x0.onehot = source if source is not None else LabelSource(labels, inst, T, label_nc)
        return x0, Act(pooled, T * per), mask

    def encode_labels(self, labels, inst, T, H, W, label_nc, fg_labels, want_mask, chunk_stride=False, source=None):
        """chunk_stride: pad the channel stride to a whole 128-byte K chunk (108 -> 128 channels) so that the
        narrow fine-scale 7x7 stems (cout <= 32) can run on the LDS-patch kernel (tile 60)."""
        per = label_nc + (1 if inst is not None else 0)
        out = self.empty_act(1, H, W, T * per)
        if chunk_stride:
            bke = 64 if self.dtype == L.BF16 else 32
            cs = (T * per + bke - 1) // bke * bke
            wide = torch.empty((1, H, W, cs), dtype=self.tdtype, device=self.device)
            self._keep(wide)
            out = Act(wide, T * per)
        mask = self.empty_f32(1, 1, H, W) if want_mask else None
        fg = None
        if want_mask:
            fg = self._fg_labels(fg_labels)
            self._keep(fg)
        if labels.dtype == torch.uint8:              # uint8 label map + int32 instance map
            if inst is not None and inst.dtype != torch.int32:
                raise TypeError("uint8 label maps go with int32 instance maps")
            fn = lib.v2v_encode_labels_u8
        else:
            if labels.dtype != torch.float32 or (inst is not None and inst.dtype != torch.float32):
                raise TypeError("label / instance maps must be fp32-encoded integers, or uint8 + int32")
            fn = lib.v2v_encode_labels
        check(fn(_ptr(labels), _ptr(inst), _ptr(out.t), _ptr(mask), T, H, W, label_nc,
                 out.Cs, _ptr(fg), 0 if fg is None else fg.numel(), self.dtype, _stream()),
              "encode_labels")
        self.label("encode_labels")
        out.onehot = source if source is not None else LabelSource(labels, inst, T, label_nc)
        return out, mask

    def widen(self, x, stride):
        """Copy an activation into a zero-padded buffer with a larger channel stride (a whole 128-byte K chunk), so
        that a 7x7 head behind a 32-channel fine-scale tower can use the LDS-patch kernel."""
        wide = torch.zeros((x.N, x.H, x.W, stride), dtype=self.tdtype, device=self.device)
        self._keep(wide)
        check(lib.v2v_concat_channels_nhwc(_ptr(x.t), x.Cs, 0, _ptr(wide), stride, 0, x.C, x.N * x.H * x.W,
                                           self.dtype, _stream()), "widen")
        self.label("widen_nhwc")
        return Act(wide, x.C)

    def concat(self, a, b):
        """torch.cat([a, b], dim=1) on NHWC activations (inference / plan path; the pad channels of the result are zero)."""
        if (a.N, a.H, a.W) != (b.N, b.H, b.W):
            raise RuntimeError("concat: spatial shapes differ")
        if self._training() and (a.t.requires_grad or b.t.requires_grad):
            raise NotImplementedError("concat has no backward: the feature-encoding first-frame generators are inference-only")
        out = self.empty_act(a.N, a.H, a.W, a.C + b.C)
        if out.Cs != a.C + b.C:
            out.t.zero_()
        P = a.N * a.H * a.W
        check(lib.v2v_concat_channels_nhwc(_ptr(a.t), a.Cs, 0, _ptr(out.t), out.Cs, 0, a.C, P, self.dtype, _stream()), "concat")
        self.label("concat_nhwc")
        check(lib.v2v_concat_channels_nhwc(_ptr(b.t), b.Cs, 0, _ptr(out.t), out.Cs, a.C, b.C, P, self.dtype, _stream()), "concat")
        self.label("concat_nhwc")
        return out

    def pack(self, x_nchw):
        N, Cc, H, W = x_nchw.shape
        if self._training() and x_nchw.requires_grad:
            from . import autograd as AG
            return Act(AG.PackFn.apply(self, x_nchw, None, 1.0), Cc)
        out = self.empty_act(N, H, W, Cc)
        check(lib.v2v_pack_nchw_to_nhwc(_ptr(x_nchw), _ptr(out.t), N, Cc, H, W, out.Cs, self.dtype, _stream()), "pack")
        self.label("pack_nchw_to_nhwc")
        return out

    def unpack(self, x):
        if self._training() and x.t.requires_grad:
            from . import autograd as AG
            return AG.UnpackFn.apply(self, x.t, x.C)
        out = self.empty_f32(x.N, x.C, x.H, x.W)
        check(lib.v2v_unpack_nhwc_to_nchw(_ptr(x.t), _ptr(out), x.N, x.C, x.H, x.W, x.Cs, self.dtype, _stream()), "unpack")
        self.label("unpack_nhwc_to_nchw")
        return out

    def avgpool_nhwc(self, x):
        OH, OW = (x.H - 1) // 2 + 1, (x.W - 1) // 2 + 1
        if self._training() and x.t.requires_grad:
            from . import autograd as AG
            return Act(AG.AvgPoolFn.apply(self, x.t), x.C)
        out = self.empty_act(x.N, OH, OW, x.C)
        if out.Cs != x.Cs:                         # the kernel has ONE channel stride: keep the input's (chunk-padded labels)
            t = torch.empty((x.N, OH, OW, x.Cs), dtype=self.tdtype, device=self.device)
            self._keep(t)
            out = Act(t, x.C)
        check(lib.v2v_avgpool3s2_nhwc(_ptr(x.t), _ptr(out.t), x.N, x.H, x.W, x.Cs, self.dtype, _stream()), "avgpool_nhwc")
        self.label("avgpool3s2_nhwc")
        return out

    def maxpool2_nhwc(self, x):
        """MaxPool2d(2, 2) (VGG19 features inside Vgg19, models/networks.py:840-870)."""
        if self._training() and x.t.requires_grad:
            from . import autograd as AG
            return Act(AG.MaxPool2Fn.apply(self, x.t), x.C)
        t = torch.empty((x.N, x.H // 2, x.W // 2, x.Cs), dtype=self.tdtype, device=self.device)
        self._keep(t)
        check(lib.v2v_maxpool2_nhwc(_ptr(x.t), _ptr(t), x.N, x.H, x.W, x.Cs, self.dtype, _stream()), "maxpool2_nhwc")
        self.label("maxpool2_nhwc")
        return Act(t, x.C)

    def avgpool2_planar(self, x):
        """AvgPool2d(2, stride 2, count_include_pad=False) on planar fp32 [..., H, W] (VGGLoss.downsample)."""
        if self._training() and x.requires_grad:
            from . import autograd as AG
            return AG.AvgPool2PlanarFn.apply(self, x)
        x = x.contiguous().float()
        H, W = x.shape[-2], x.shape[-1]
        out = self.empty_f32(*x.shape[:-2], H // 2, W // 2)
        check(lib.v2v_avgpool2_planar(_ptr(x), _ptr(out), x.numel() // (H * W), H, W, _stream()), "avgpool2_planar")
        self.label("avgpool2_planar")
        return out

    def instance_mean(self, feat, inst):
        """Instance-wise average pooling (Encoder.forward, models/networks.py:621-632).  feat: planar fp32 (N, C, H, W);
        inst: (N, 1, H, W) fp32 holding integer ids.  Returns a tensor like feat."""
        feat = feat.contiguous().float()
        inst = inst.contiguous().float()
        N, Cc, H, W = feat.shape
        out = self.empty_f32(N, Cc, H, W)
        nbytes = lib.v2v_instance_mean_workspace(Cc, H * W)
        ws = self.scratch("instance_mean_ws", (nbytes + 3) // 4)
        for n in range(N):
            check(lib.v2v_instance_mean_planar(_ptr(feat[n]), _ptr(inst[n]), _ptr(out[n]), _ptr(ws), Cc, H * W, _stream()),
                  "instance_mean")
            self.label("instance_mean")
        return out

    def onehot_planar(self, labels, inst, H, W, label_nc):
        """Planar fp32 one-hot (+ edge plane) of one label frame: `real_A[0][0, -1]` (vid2vid_model_G.py:209)."""
        per = label_nc + (1 if inst is not None else 0)
        out = self.empty_f32(per, H, W)
        fn = lib.v2v_onehot_planar_u8 if labels.dtype == torch.uint8 else lib.v2v_onehot_planar
        check(fn(_ptr(labels), _ptr(inst), _ptr(out), H, W, label_nc, _stream()), "onehot_planar")
        self.label("onehot_planar")
        return out

    def avgpool_planar(self, x):
        """x: fp32 [..., H, W] contiguous -> [..., OH, OW]"""
        H, W = x.shape[-2], x.shape[-1]
        OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        planes = x.numel() // (H * W)
        out = self.empty_f32(*x.shape[:-2], OH, OW)
        check(lib.v2v_avgpool3s2_planar(_ptr(x), _ptr(out), planes, H, W, _stream()), "avgpool_planar")
        self.label("avgpool3s2_planar")
        return out

    def warp_blend(self, img_raw, flow, weight, prev, fg, mask, want_warp=False):
        N, Cc, H, W = img_raw.shape
        if self._training() and any(t is not None and t.requires_grad for t in (img_raw, flow, weight, prev, fg)):
            from . import autograd as AG
            final, raw_blend = AG.WarpBlendFn.apply(self, img_raw, flow, weight, prev, fg, mask)
            return (final, raw_blend), None
        gx, gy = self.grid(H, W)
        final = self.empty_f32(N, Cc, H, W)
        warp = self.empty_f32(N, Cc, H, W) if (want_warp and flow is not None) else None
        for t in (gx, gy):
            self._keep(t)
        check(lib.v2v_warp_blend(_ptr(img_raw), _ptr(flow), _ptr(weight), _ptr(prev), _ptr(fg), _ptr(mask),
                                 _ptr(final), _ptr(warp), _ptr(gx), _ptr(gy), N, Cc, H, W,
                                 int(self.align_corners), _stream()), "warp_blend")
        self.label("warp_blend")
        return final, warp

    def resample_flow(self, img, flow):
        N, Cc, H, W = img.shape
        if self._training() and (img.requires_grad or flow.requires_grad):
            from . import autograd as AG
            return AG.ResampleFn.apply(self, img, flow)
        gx, gy = self.grid(H, W)
        out = self.empty_f32(N, Cc, H, W)
        check(lib.v2v_resample_flow(_ptr(img), _ptr(flow), _ptr(out), _ptr(gx), _ptr(gy), N, Cc, H, W,
                                    int(self.align_corners), _stream()), "resample_flow")
        self.label("resample_flow")
        return out
