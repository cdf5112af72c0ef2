"""vid2vid network zoo on the MI355X backend.

Drop-in for the reference's models/networks.py: same factory functions (`define_G`,
`define_D`), same class names and constructor signatures, and -- because checkpoints are
the on-disk contract (models/base_model.py:43-107) -- the same parameter names and shapes:
every network is assembled from the same nn.Sequential index layout as the reference
(models/networks.py:117-201, :234-294, :327-353, :361-400, :554-589, :634-715), with
torch.nn modules used purely as parameter containers.  Module creation order also follows
the reference so that a seeded construction draws identical initial weights.

`forward()` never runs torch.nn compute: it lowers the module lists to libv2v_hip.so
launches through `engine.Engine` (implicit-GEMM MFMA convolutions with fused padding,
training-mode norm, activations, residuals; fused warp-and-blend tail).
"""
import copy
import functools

import numpy as np
import torch
import torch.nn as nn

from . import lib as L
from .engine import Engine, Act


# --------------------------------------------------------------------------------------
# engine registry: one Engine per (device, precision)
# --------------------------------------------------------------------------------------
_PRECISION = {"value": L.F32}
_ALIGN_CORNERS = {"value": False}   # oracle in this container: torch>=1.3 default
_ENGINES = {}


X3 = L.F32 + 16          # precision code of the fp32 engine with bf16x3 operands for its 3x3 convolutions (engine.X3Conv)


def set_precision(p):
    """'fp32' (exact fp32 MFMA, parity path), 'bf16' (throughput path) or 'x3' (fp32 storage / norms, the 3x3 convolutions on
    the bf16 matrix pipe over bf16x3 operands: fp32-grade results at a multiple of the fp32 path's speed)."""
    _PRECISION["value"] = {"fp32": L.F32, "f32": L.F32, "bf16": L.BF16, "x3": X3}[p]


def get_precision():
    return {L.BF16: "bf16", X3: "x3"}.get(_PRECISION["value"], "fp32")


def set_align_corners(flag):
    """grid_sample convention: False = current torch default (what the reference executes
    under torch>=1.3), True = PyTorch-0.4 behaviour the checkpoints were trained with."""
    _ALIGN_CORNERS["value"] = bool(flag)


_RECORD_ONLY = {"value": False}


def set_record_only(flag):
    """CPU dry-run mode for the test-suite: engines on a CPU device may RECORD plans (argument
    validation, layer census) but can never execute anything.  Not a compute fallback."""
    _RECORD_ONLY["value"] = bool(flag)
    L.lib.v2v_set_dry_run(1 if flag else 0)


# ---- "these inputs are ready" events (round 6) ----------------------------------------------------------------------------
# Vid2VidModelG.forward records an event right behind its input handling (the real frames are on the device from there on) and files
# it under the frames' storage; FlowNet.forward, which train.py calls AFTER modelG(...) returns but whose inputs are exactly those
# frames, then waits for that event instead of for everything the generator has enqueued since -- so FlowNet2 runs beside the
# generator's forward pass on its own stream (models/flownet.py).  Unknown storages fall back to a full stream wait.
_READY = {}


def note_inputs_ready(t):
    if t is None or not t.is_cuda:
        return
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(t.device))
    if len(_READY) >= 8:
        _READY.pop(next(iter(_READY)))
    # the tensor itself is kept (a few frames, at most 8 entries): while it is alive its storage cannot be handed to another tensor,
    # so a later lookup by address can never find the event of a dead tensor's memory
    _READY[(t.device.index, t.untyped_storage().data_ptr())] = (ev, t)


def inputs_ready_event(t):
    ent = _READY.get((t.device.index, t.untyped_storage().data_ptr())) if t.is_cuda else None
    return None if ent is None else ent[0]


def get_engine(device=None, precision=None):
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
    device = torch.device(device)
    if device.type == "cuda" and device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    prec = _PRECISION["value"] if precision is None else precision
    key = (str(device), prec, _ALIGN_CORNERS["value"])
    if key not in _ENGINES:
        _ENGINES[key] = Engine(device, prec & 15, _ALIGN_CORNERS["value"],
                               record_only=_RECORD_ONLY["value"] and device.type == "cpu", x3=bool(prec & 16))
    return _ENGINES[key]


_PREC_CODE = {"fp32": L.F32, "f32": L.F32, "bf16": L.BF16, "x3": X3}


def bind_precision(root, precision):
    """Pin every sub-module of `root` (a model or a network) to ONE precision, so that two models of different
    precision can live in one process (bench.py checks the bf16 model against an fp32 model and the oracle):
    `module_engine` then resolves the engine from the module instead of the process-wide default."""
    code = _PREC_CODE[precision] if isinstance(precision, str) else precision
    for m in root.modules():
        m._v2v_precision = code


def module_engine(mod, device):
    """Engine of (device, the precision `mod` was bound to | the process-wide default)."""
    return get_engine(device, getattr(mod, "_v2v_precision", None))


# --------------------------------------------------------------------------------------
# init / factories  (models/networks.py:15-68)
# --------------------------------------------------------------------------------------
def weights_init(m):
    kind = type(m).__name__
    if "Conv" in kind and hasattr(m, "weight"):
        m.weight.data.normal_(0.0, 0.02)
    elif "BatchNorm2d" in kind:
        m.weight.data.normal_(1.0, 0.02)
        m.bias.data.fill_(0)


def get_norm_layer(norm_type="instance"):
    if norm_type == "batch":
        return functools.partial(nn.BatchNorm2d, affine=True)
    if norm_type == "instance":
        return functools.partial(nn.InstanceNorm2d, affine=False, track_running_stats=True)
    raise NotImplementedError("normalization layer [%s] is not found" % norm_type)


def define_G(input_nc, output_nc, prev_output_nc, ngf, which_model_netG, n_downsampling, norm, scale,
             gpu_ids=[], opt=[]):
    norm_layer = get_norm_layer(norm_type=norm)
    if which_model_netG == "global":
        netG = GlobalGenerator(input_nc, output_nc, ngf, n_downsampling, opt.n_blocks, norm_layer)
    elif which_model_netG == "local":
        netG = LocalEnhancer(input_nc, output_nc, ngf, n_downsampling, opt.n_blocks, opt.n_local_enhancers,
                             opt.n_blocks_local, norm_layer)
    elif which_model_netG == "composite":
        netG = CompositeGenerator(opt, input_nc, output_nc, prev_output_nc, ngf, n_downsampling, opt.n_blocks,
                                  opt.fg, opt.no_flow, norm_layer)
    elif which_model_netG == "compositeLocal":
        netG = CompositeLocalGenerator(opt, input_nc, output_nc, prev_output_nc, ngf, n_downsampling,
                                       opt.n_blocks_local, opt.fg, opt.no_flow, norm_layer, scale=scale)
    elif which_model_netG == "global_with_features":
        netG = Global_with_z(input_nc, output_nc, opt.feat_num, ngf, n_downsampling, opt.n_blocks, norm_layer)
    elif which_model_netG == "local_with_features":
        netG = Local_with_z(input_nc, output_nc, opt.feat_num, ngf, n_downsampling, opt.n_blocks, opt.n_local_enhancers,
                            opt.n_blocks_local, norm_layer)
    elif which_model_netG == "encoder":
        netG = Encoder(input_nc, output_nc, ngf, n_downsampling, norm_layer)
    else:
        raise NotImplementedError("Generator model name [%s] is not recognized" % which_model_netG)
    if len(gpu_ids) > 0 and gpu_ids[0] >= 0:
        netG.cuda(gpu_ids[0])
    netG.apply(weights_init)
    return netG


def define_D(input_nc, ndf, n_layers_D, norm="instance", num_D=1, getIntermFeat=False, gpu_ids=[]):
    norm_layer = get_norm_layer(norm_type=norm)
    netD = MultiscaleDiscriminator(input_nc, ndf, n_layers_D, norm_layer, num_D, getIntermFeat)
    if len(gpu_ids) > 0 and gpu_ids[0] >= 0:
        netD.cuda(gpu_ids[0])
    netD.apply(weights_init)
    return netD


def print_network(net):
    if isinstance(net, list):
        net = net[0]
    print(net)
    print("Total number of parameters: %d" % sum(p.numel() for p in net.parameters()))


# --------------------------------------------------------------------------------------
# layer-list builders (index layout == reference's, which fixes the state_dict keys)
# --------------------------------------------------------------------------------------
def _stem7(cin, cout, norm_layer):
    return [nn.ReflectionPad2d(3), nn.Conv2d(cin, cout, kernel_size=7, padding=0), norm_layer(cout), nn.ReLU(True)]


def _down3(cin, cout, norm_layer):
    return [nn.Conv2d(cin, cout, kernel_size=3, stride=2, padding=1), norm_layer(cout), nn.ReLU(True)]


def _up3(cin, cout, norm_layer):
    return [nn.ConvTranspose2d(cin, cout, kernel_size=3, stride=2, padding=1, output_padding=1),
            norm_layer(cout), nn.ReLU(True)]


def _head7(cin, cout, final_act=None):
    mods = [nn.ReflectionPad2d(3), nn.Conv2d(cin, cout, kernel_size=7, padding=0)]
    if final_act is not None:
        mods.append(final_act)
    return mods


class ResnetBlock(nn.Module):
    """x + norm(conv3(pad(act(norm(conv3(pad(x)))))))   (models/networks.py:554-593)"""

    def __init__(self, dim, padding_type, norm_layer, activation=nn.ReLU(True), use_dropout=False):
        super().__init__()
        if use_dropout:
            raise NotImplementedError("dropout is never enabled by vid2vid")
        if padding_type not in ("reflect", "zero"):
            raise NotImplementedError("padding [%s] is not implemented" % padding_type)
        p = 1 if padding_type == "zero" else 0
        blk = []
        for half in range(2):
            if padding_type == "reflect":
                blk.append(nn.ReflectionPad2d(1))
            blk += [nn.Conv2d(dim, dim, kernel_size=3, padding=p), norm_layer(dim)]
            if half == 0:
                blk.append(activation)
        self.conv_block = nn.Sequential(*blk)

    def forward(self, x):
        eng = module_engine(self, x.device)
        a = eng.pack(x.contiguous().float())
        return eng.unpack(eng.run_resblock(self, a, None, "resblock"))


class BaseNetwork(nn.Module):
    def _engine(self, ref):
        dev = ref.t.device if isinstance(ref, Act) else ref.device
        return module_engine(self, dev)

    @staticmethod
    def _as_act(eng, x):
        if x is None or isinstance(x, Act):
            return x
        return eng.pack(x.contiguous().float())

    def resample(self, image, flow):
        """BaseNetwork.resample (models/networks.py:108-115) on planar fp32 tensors."""
        eng = module_engine(self, image.device)
        return eng.resample_flow(image.contiguous().float(), flow.contiguous().float())

    def _tail(self, eng, img_raw, flow, weight, img_prev_nchw, img_fg, mask, use_raw_only):
        """models/networks.py:215-230 / :309-323 as one fused launch."""
        do_warp = not (use_raw_only or self.no_flow)
        if not do_warp and img_fg is None:
            return img_raw, img_raw
        prev3 = img_prev_nchw[:, -3:].contiguous() if do_warp else None
        if mask is not None:
            mask = mask.contiguous().float()
        res, _ = eng.warp_blend(img_raw, flow if do_warp else None, weight if do_warp else None, prev3,
                                img_fg, mask if img_fg is not None else None)
        if isinstance(res, tuple):          # training graph: (img_final, blended img_raw), inputs untouched
            return res
        return res, img_raw                 # inference: img_raw was blended in place


class CompositeGenerator(BaseNetwork):
    """Coarsest-scale generator (models/networks.py:117-232)."""

    def __init__(self, opt, input_nc, output_nc, prev_output_nc, ngf, n_downsampling, n_blocks, use_fg_model=False,
                 no_flow=False, norm_layer=nn.BatchNorm2d, padding_type="reflect"):
        assert n_blocks >= 0
        super().__init__()
        self.opt = opt
        self.n_downsampling = n_downsampling
        self.use_fg_model = use_fg_model
        self.no_flow = no_flow
        act = nn.ReLU(True)
        res = lambda c: ResnetBlock(c, padding_type=padding_type, activation=act, norm_layer=norm_layer)

        if use_fg_model:
            nf = ngf // 2 if n_downsampling > 2 else ngf
            indv_down = _stem7(input_nc, nf, norm_layer)
            indv_down[-1] = act
            for i in range(n_downsampling):
                indv_down += _down3(nf * 2 ** i, nf * 2 ** (i + 1), norm_layer)[:2] + [act]
            indv_res = [res(nf * 2 ** n_downsampling) for _ in range(n_blocks)]
            indv_up = []
            for i in range(n_downsampling):
                m = 2 ** (n_downsampling - i)
                indv_up += _up3(nf * m, nf * m // 2, norm_layer)[:2] + [act]
            indv_final = _head7(nf, output_nc, nn.Tanh())

        down_seg = _stem7(input_nc, ngf, norm_layer)
        down_seg[-1] = act
        for i in range(n_downsampling):
            down_seg += _down3(ngf * 2 ** i, ngf * 2 ** (i + 1), norm_layer)[:2] + [act]
        top = ngf * 2 ** n_downsampling
        down_seg += [res(top) for _ in range(n_blocks - n_blocks // 2)]
        down_img = _stem7(prev_output_nc, ngf, norm_layer)
        down_img[-1] = act
        down_img += copy.deepcopy(down_seg[4:])

        res_img = [res(top) for _ in range(n_blocks // 2)]
        if not no_flow:
            res_flow = copy.deepcopy(res_img)
        up_img = []
        for i in range(n_downsampling):
            m = 2 ** (n_downsampling - i)
            up_img += _up3(ngf * m, ngf * m // 2, norm_layer)[:2] + [act]
        final_img = _head7(ngf, output_nc, nn.Tanh())
        if not no_flow:
            up_flow = copy.deepcopy(up_img)
            final_flow = _head7(ngf, 2)
            final_w = _head7(ngf, 1, nn.Sigmoid())

        if use_fg_model:
            self.indv_down = nn.Sequential(*indv_down)
            self.indv_res = nn.Sequential(*indv_res)
            self.indv_up = nn.Sequential(*indv_up)
            self.indv_final = nn.Sequential(*indv_final)
        self.model_down_seg = nn.Sequential(*down_seg)
        self.model_down_img = nn.Sequential(*down_img)
        self.model_res_img = nn.Sequential(*res_img)
        self.model_up_img = nn.Sequential(*up_img)
        self.model_final_img = nn.Sequential(*final_img)
        if not no_flow:
            self.model_res_flow = nn.Sequential(*res_flow)
            self.model_up_flow = nn.Sequential(*up_flow)
            self.model_final_flow = nn.Sequential(*final_flow)
            self.model_final_w = nn.Sequential(*final_w)

    def flow_multiplier(self):
        return 20.0

    def label_stems(self):
        """The convolutions that read the encoded label maps directly (7x7 stems of the label and foreground towers)."""
        convs = [next(mm for mm in self.model_down_seg if isinstance(mm, nn.Conv2d))]
        if self.use_fg_model:
            convs.append(next(mm for mm in self.indv_down if isinstance(mm, nn.Conv2d)))
        return convs

    def emit(self, eng, x, prev, img_prev_nchw, mask, img_feat_coarse, flow_feat_coarse, img_fg_feat_coarse,
             use_raw_only, tag="G0"):
        """x: Act labels (NHWC), prev: Act previous frames (NHWC), img_prev_nchw: fp32 planar.
        The label tower, the image tower and the foreground tower are independent until they are summed / blended, and
        so are the image and flow branches behind the sum (models/networks.py:203-232): with `eng.lanes_enabled` (frame
        plans) they are emitted on parallel plan lanes = parallel hipGraph paths."""
        lanes = eng.lanes_enabled
        img_fg = img_fg_feat = None

        def fg_tower():
            f = eng.run_sequential(self.indv_down, x, name=tag + ".indv_down")
            f = eng.run_sequential(self.indv_res, f, name=tag + ".indv_res")
            feat = eng.run_sequential(self.indv_up, f, name=tag + ".indv_up")
            return feat, eng.run_sequential(self.indv_final, feat, head_nchw=True, name=tag + ".indv_final")

        def flow_heads(flow_feat):
            both = eng.head_pair(flow_feat, self.model_final_flow, self.flow_multiplier(), self.model_final_w, 1.0,
                                 label=tag + ".final_flow+w")        # the flow branch ends the frame's critical path: one launch
            if both is not None:
                return both
            flow = eng.run_sequential(self.model_final_flow, flow_feat, head_nchw=True,
                                      out_scale=self.flow_multiplier(), name=tag + ".final_flow")
            return flow, eng.run_sequential(self.model_final_w, flow_feat, head_nchw=True, name=tag + ".final_w")

        def flow_branch(down):
            res_flow = eng.run_sequential(self.model_res_flow, down, name=tag + ".res_flow")
            flow_feat = eng.run_sequential(self.model_up_flow, res_flow, name=tag + ".up_flow")
            flow, weight = flow_heads(flow_feat)
            return flow, weight, flow_feat

        twin = eng.twin_enabled and not eng._training()
        res_flow_done = None
        if twin:
            # twin chains as paired launches (engine.run_resblocks_twin): the towers' stems / down-convs stay on their own
            # lanes, their ResnetBlock chains (identical shapes, different weights) advance in lock step on lane 0, and so do
            # model_res_img / model_res_flow, which both start from the tower sum
            def split(seq):
                mods = list(seq)
                k = next((i for i, m in enumerate(mods) if hasattr(m, "conv_block")), len(mods))
                return mods[:k], mods[k:]
            head_seg, blocks_seg = split(self.model_down_seg)
            head_img, blocks_img = split(self.model_down_img)
            with eng.on_lane(1):
                # the label stem (108 -> 128, 7x7 at full resolution: 178 GFLOP) heads the frame's critical path: the
                # foreground tower (far off that path) forks only behind it instead of competing with it for the chip
                k_stem = next((i + 1 for i, m in enumerate(head_seg) if isinstance(m, nn.ReLU)), len(head_seg))
                seg = eng.run_sequential(head_seg[:k_stem], x, name=tag + ".down_seg")
                if self.use_fg_model and lanes:
                    with eng.on_lane(2):
                        img_fg_feat, img_fg = fg_tower()
                seg = eng.run_sequential(head_seg[k_stem:], seg, name=tag + ".down_seg", first_index=k_stem)
            img = eng.run_sequential(head_img, prev, name=tag + ".down_img")
            eng.join(1)
            seg, img = eng.run_resblocks_twin(blocks_seg, seg, blocks_img, img, tag + ".down_seg", tag + ".down_img",
                                              first_index=len(head_seg))
            down = eng.add(img, seg)
        elif lanes:
            with eng.on_lane(1):
                seg = eng.run_sequential(self.model_down_seg, x, name=tag + ".down_seg")
            if self.use_fg_model:
                with eng.on_lane(2):
                    img_fg_feat, img_fg = fg_tower()
            down = eng.run_sequential(self.model_down_img, prev, name=tag + ".down_img")
            eng.join(1)
            down = eng.add(down, seg)                    # the tower sum as its own launch (fused into the last norm otherwise)
        else:
            seg = eng.run_sequential(self.model_down_seg, x, name=tag + ".down_seg")
            down = eng.run_sequential(self.model_down_img, prev, extra_add=seg, name=tag + ".down_img")
        flow = weight = flow_feat = None
        res_img_out = None
        if twin and not self.no_flow:
            res_img_out, res_flow_done = eng.run_resblocks_twin(self.model_res_img, down, self.model_res_flow, down,
                                                                tag + ".res_img", tag + ".res_flow")

        def flow_tail(res_flow):
            flow_feat = eng.run_sequential(self.model_up_flow, res_flow, name=tag + ".up_flow")
            flow, weight = flow_heads(flow_feat)
            return flow, weight, flow_feat

        if res_flow_done is not None:
            with eng.on_lane(1):
                flow, weight, flow_feat = flow_tail(res_flow_done)
        elif lanes and not self.no_flow:
            with eng.on_lane(1):
                flow, weight, flow_feat = flow_branch(down)
        if res_img_out is None:
            res_img_out = eng.run_sequential(self.model_res_img, down, name=tag + ".res_img")
        img_feat = eng.run_sequential(self.model_up_img, res_img_out, name=tag + ".up_img")
        img_raw = eng.run_sequential(self.model_final_img, img_feat, head_nchw=True, name=tag + ".final_img")
        if twin and self.use_fg_model and not lanes:
            img_fg_feat, img_fg = fg_tower()
        if not lanes and not self.no_flow and flow is None:
            flow, weight, flow_feat = flow_branch(down)
        if not lanes and self.use_fg_model and img_fg is None:
            img_fg_feat, img_fg = fg_tower()
        if lanes or twin:
            eng.join(1)
            if self.use_fg_model:
                eng.join(2)
        img_final, img_raw = self._tail(eng, img_raw, flow, weight, img_prev_nchw, img_fg, mask, use_raw_only)
        return img_final, flow, weight, img_raw, img_feat, flow_feat, img_fg_feat

    def forward(self, input, img_prev, mask, img_feat_coarse, flow_feat_coarse, img_fg_feat_coarse, use_raw_only):
        """Same signature / return tuple as the reference (models/networks.py:203-232).  `input`,
        `img_prev`, `mask` are NCHW fp32 device tensors; feature maps are returned as NHWC `Act`
        handles (they only ever feed the next scale's forward)."""
        eng = module_engine(self, input.device)
        img_prev = img_prev.contiguous().float()
        return self.emit(eng, self._as_act(eng, input), eng.pack(img_prev), img_prev, mask,
                         img_feat_coarse, flow_feat_coarse, img_fg_feat_coarse, use_raw_only)


class CompositeLocalGenerator(BaseNetwork):
    """Finer-scale generator (models/networks.py:234-325)."""

    def __init__(self, opt, input_nc, output_nc, prev_output_nc, ngf, n_downsampling, n_blocks_local,
                 use_fg_model=False, no_flow=False, norm_layer=nn.BatchNorm2d, padding_type="reflect", scale=1):
        super().__init__()
        self.opt = opt
        self.use_fg_model = use_fg_model
        self.no_flow = no_flow
        self.scale = scale
        act = nn.ReLU(True)
        res = lambda c: ResnetBlock(c, padding_type=padding_type, activation=act, norm_layer=norm_layer)

        def stem_down(cin, nf):
            m = _stem7(cin, nf, norm_layer)
            m[-1] = act
            return m + _down3(nf, nf * 2, norm_layer)[:2] + [act]

        if use_fg_model:
            nf = ngf // 2 if n_downsampling > 2 else ngf
            indv_down = stem_down(input_nc, nf)
            indv_up = [res(nf * 2) for _ in range(n_blocks_local)]
            indv_up += _up3(nf * 2, nf, norm_layer)[:2] + [act]
            indv_final = _head7(nf, output_nc, nn.Tanh())

        down_seg = stem_down(input_nc, ngf)
        down_img = stem_down(prev_output_nc, ngf)
        up_img = [res(ngf * 2) for _ in range(n_blocks_local)]
        up_img += _up3(ngf * 2, ngf, norm_layer)[:2] + [act]
        final_img = _head7(ngf, output_nc, nn.Tanh())
        if not no_flow:
            up_flow = copy.deepcopy(up_img)
            final_flow = _head7(ngf, 2)
            final_w = _head7(ngf, 1, nn.Sigmoid())

        if use_fg_model:
            self.indv_down = nn.Sequential(*indv_down)
            self.indv_up = nn.Sequential(*indv_up)
            self.indv_final = nn.Sequential(*indv_final)
        self.model_down_seg = nn.Sequential(*down_seg)
        self.model_down_img = nn.Sequential(*down_img)
        self.model_up_img = nn.Sequential(*up_img)
        self.model_final_img = nn.Sequential(*final_img)
        if not no_flow:
            self.model_up_flow = nn.Sequential(*up_flow)
            self.model_final_flow = nn.Sequential(*final_flow)
            self.model_final_w = nn.Sequential(*final_w)

    def flow_multiplier(self):
        return 20.0 * (2 ** self.scale)

    def label_stems(self):
        """The convolutions that read the encoded label maps directly (7x7 stems of the label and foreground towers): when all
        of them run as gather-sums on the maps, the full-resolution encoding is never written (Engine.encode_labels_pooled)."""
        convs = [next(mm for mm in self.model_down_seg if isinstance(mm, nn.Conv2d))]
        if self.use_fg_model:
            convs.append(next(mm for mm in self.indv_down if isinstance(mm, nn.Conv2d)))
        return convs

    def emit(self, eng, x, prev, img_prev_nchw, mask, img_feat_coarse, flow_feat_coarse, img_fg_feat_coarse,
             use_raw_only, tag="G1"):
        """Same lane structure as CompositeGenerator.emit: label stem | image stem | foreground branch in parallel, then
        image branch | flow branch (models/networks.py:296-325)."""
        lanes = eng.lanes_enabled
        img_fg = img_fg_feat = None

        def fg_branch():
            f = eng.run_sequential(self.indv_down, x, extra_add=img_fg_feat_coarse, name=tag + ".indv_down")
            feat = eng.run_sequential(self.indv_up, f, name=tag + ".indv_up")
            return feat, eng.run_sequential(self.indv_final, feat, head_nchw=True, name=tag + ".indv_final")

        def flow_branch(down):
            flow_feat = eng.run_sequential(self.model_up_flow, eng.add(down, flow_feat_coarse), name=tag + ".up_flow")
            both = eng.head_pair(flow_feat, self.model_final_flow, self.flow_multiplier(), self.model_final_w, 1.0,
                                 label=tag + ".final_flow+w")
            if both is not None:
                return both[0], both[1], flow_feat
            flow = eng.run_sequential(self.model_final_flow, flow_feat, head_nchw=True,
                                      out_scale=self.flow_multiplier(), name=tag + ".final_flow")
            weight = eng.run_sequential(self.model_final_w, flow_feat, head_nchw=True, name=tag + ".final_w")
            return flow, weight, flow_feat

        if lanes:
            with eng.on_lane(1):
                seg = eng.run_sequential(self.model_down_seg, x, name=tag + ".down_seg")
            if self.use_fg_model:
                with eng.on_lane(2):
                    img_fg_feat, img_fg = fg_branch()
            down = eng.run_sequential(self.model_down_img, prev, name=tag + ".down_img")
            eng.join(1)
            down = eng.add(down, seg)
        else:
            seg = eng.run_sequential(self.model_down_seg, x, name=tag + ".down_seg")
            down = eng.run_sequential(self.model_down_img, prev, extra_add=seg, name=tag + ".down_img")
        flow = weight = flow_feat = None
        if lanes and not self.no_flow:
            with eng.on_lane(1):
                flow, weight, flow_feat = flow_branch(down)
        img_feat = eng.run_sequential(self.model_up_img, eng.add(down, img_feat_coarse), name=tag + ".up_img")
        img_raw = eng.run_sequential(self.model_final_img, img_feat, head_nchw=True, name=tag + ".final_img")
        if not lanes and not self.no_flow:
            flow, weight, flow_feat = flow_branch(down)
        if not lanes and self.use_fg_model:
            img_fg_feat, img_fg = fg_branch()
        if lanes:
            eng.join(1)
            if self.use_fg_model:
                eng.join(2)
        img_final, img_raw = self._tail(eng, img_raw, flow, weight, img_prev_nchw, img_fg, mask, use_raw_only)
        return img_final, flow, weight, img_raw, img_feat, flow_feat, img_fg_feat

    def forward(self, input, img_prev, mask, img_feat_coarse, flow_feat_coarse, img_fg_feat_coarse, use_raw_only):
        eng = module_engine(self, input.device)
        img_prev = img_prev.contiguous().float()
        return self.emit(eng, self._as_act(eng, input), eng.pack(img_prev), img_prev, mask,
                         img_feat_coarse, flow_feat_coarse, img_fg_feat_coarse, use_raw_only)


class GlobalGenerator(nn.Module):
    """pix2pixHD global generator used for the first frame (models/networks.py:327-359)."""

    def __init__(self, input_nc, output_nc, ngf=64, n_downsampling=3, n_blocks=9, norm_layer=nn.BatchNorm2d,
                 padding_type="reflect"):
        assert n_blocks >= 0
        super().__init__()
        act = nn.ReLU(True)
        cap = lambda c: min(1024, c)
        model = _stem7(input_nc, ngf, norm_layer)
        model[-1] = act
        for i in range(n_downsampling):
            model += _down3(cap(ngf * 2 ** i), cap(ngf * 2 ** (i + 1)), norm_layer)[:2] + [act]
        top = cap(ngf * 2 ** n_downsampling)
        model += [ResnetBlock(top, padding_type=padding_type, activation=act, norm_layer=norm_layer)
                  for _ in range(n_blocks)]
        for i in range(n_downsampling):
            m = 2 ** (n_downsampling - i)
            model += _up3(cap(ngf * m), cap(int(ngf * m / 2)), norm_layer)[:2] + [act]
        model += _head7(ngf, output_nc, nn.Tanh())
        self.model = nn.Sequential(*model)

    def emit(self, eng, x):
        return eng.run_sequential(self.model, x, head_nchw=True, name="Gi")

    def forward(self, input, feat=None):
        if feat is not None:
            input = torch.cat([input, feat], dim=1)
        eng = module_engine(self, input.device)
        return self.emit(eng, eng.pack(input.contiguous().float()))


class LocalEnhancer(nn.Module):
    """pix2pixHD local enhancer used for the 2048-wide first frame (models/networks.py:361-419)."""

    def __init__(self, input_nc, output_nc, ngf=32, n_downsample_global=3, n_blocks_global=9, n_local_enhancers=1,
                 n_blocks_local=3, norm_layer=nn.BatchNorm2d, padding_type="reflect"):
        super().__init__()
        self.n_local_enhancers = n_local_enhancers
        g = GlobalGenerator(input_nc, output_nc, ngf * (2 ** n_local_enhancers), n_downsample_global,
                            n_blocks_global, norm_layer).model
        self.model = nn.Sequential(*[g[i] for i in range(len(g) - 3)])     # drop the output head
        for n in range(1, n_local_enhancers + 1):
            nf = ngf * (2 ** (n_local_enhancers - n))
            down = _stem7(input_nc, nf, norm_layer) + _down3(nf, nf * 2, norm_layer)
            up = [ResnetBlock(nf * 2, padding_type=padding_type, norm_layer=norm_layer) for _ in range(n_blocks_local)]
            up += _up3(nf * 2, nf, norm_layer)
            if n == n_local_enhancers:
                up += _head7(ngf, output_nc, nn.Tanh())
            setattr(self, "model%d_1" % n, nn.Sequential(*down))
            setattr(self, "model%d_2" % n, nn.Sequential(*up))
        self.downsample = nn.AvgPool2d(3, stride=2, padding=[1, 1], count_include_pad=False)

    def forward(self, input, feat_map=None):
        if feat_map is not None:
            input = torch.cat([input, feat_map], dim=1)
        eng = module_engine(self, input.device)
        return self.emit(eng, eng.pack(input.contiguous().float()))

    def emit(self, eng, x):
        pyr = [x]
        for _ in range(self.n_local_enhancers):
            pyr.append(eng.avgpool_nhwc(pyr[-1]))
        out = eng.run_sequential(self.model, pyr[-1], name="Gi.global")
        for n in range(1, self.n_local_enhancers + 1):
            last = n == self.n_local_enhancers
            d = eng.run_sequential(getattr(self, "model%d_1" % n), pyr[self.n_local_enhancers - n], extra_add=out,
                                   name="Gi.local%d_1" % n)
            out = eng.run_sequential(getattr(self, "model%d_2" % n), d, head_nchw=last, name="Gi.local%d_2" % n)
        return out


# --------------------------------------------------------------------------------------
# first-frame generators with instance-wise feature encoding (models/networks.py:421-632)
# Parameter containers + lowering are composed from the validated primitives (concat, avgpool, conv groups); parity on the
# GPU: tests/test_gpu_golden.py::test_feature_encoding_first_frame_nets_vs_reference against the reference's own classes
# (tests/golden/face_first_frame_nets_32x32.npz).
# --------------------------------------------------------------------------------------
class Global_with_z(nn.Module):
    """models/networks.py:421-467: GlobalGenerator whose stem, residual trunk, up-sampling trunk and head each see the
    feature map z (nz channels, average-pooled to the trunk's resolution) concatenated to their input."""

    def __init__(self, input_nc, output_nc, nz, ngf=64, n_downsample_G=3, n_blocks=9, norm_layer=nn.BatchNorm2d,
                 padding_type="reflect"):
        super().__init__()
        self.n_downsample_G = n_downsample_G
        act = nn.ReLU(True)
        cap = lambda c: min(1024, c)
        down = _stem7(input_nc + nz, ngf, norm_layer)
        down[-1] = act
        for i in range(n_downsample_G):
            down += _down3(cap(ngf * 2 ** i), cap(ngf * 2 ** (i + 1)), norm_layer)[:2] + [act]
        top = cap(ngf * 2 ** n_downsample_G)
        res = [ResnetBlock(top + nz, padding_type=padding_type, norm_layer=norm_layer) for _ in range(n_blocks)]
        up = []
        for i in range(n_downsample_G):
            m = 2 ** (n_downsample_G - i)
            cin = cap(ngf * m) + (nz * 2 if i == 0 else 0)
            up += _up3(cin, cap(ngf * m // 2), norm_layer)[:2] + [act]
        head = _head7(ngf + nz, output_nc, nn.Tanh())
        self.model_downsample = nn.Sequential(*down)
        self.model_resnet = nn.Sequential(*res)
        self.model_upsample = nn.Sequential(*up)
        self.model_upsample_conv = nn.Sequential(*head)
        self.downsample = nn.AvgPool2d(3, stride=2, padding=[1, 1], count_include_pad=False)

    def emit(self, eng, x, z, tag="Gz"):
        zd = z
        for _ in range(self.n_downsample_G):
            zd = eng.avgpool_nhwc(zd)
        h = eng.run_sequential(self.model_downsample, eng.concat(x, z), name=tag + ".down")
        h = eng.run_sequential(self.model_resnet, eng.concat(h, zd), name=tag + ".res")
        h = eng.run_sequential(self.model_upsample, eng.concat(h, zd), name=tag + ".up")
        return eng.run_sequential(self.model_upsample_conv, eng.concat(h, z), head_nchw=True, name=tag + ".head")

    def forward(self, x, z):
        eng = module_engine(self, x.device)
        with torch.no_grad():
            return self.emit(eng, eng.pack(x.contiguous().float()), eng.pack(z.contiguous().float()))


class Local_with_z(nn.Module):
    """models/networks.py:469-552."""

    def __init__(self, input_nc, output_nc, nz, ngf=32, n_downsample_global=3, n_blocks_global=9, n_local_enhancers=1,
                 n_blocks_local=3, norm_layer=nn.BatchNorm2d, padding_type="reflect"):
        super().__init__()
        self.n_local_enhancers = n_local_enhancers
        self.n_downsample_global = n_downsample_global
        g = Global_with_z(input_nc, output_nc, nz, ngf * (2 ** n_local_enhancers), n_downsample_global, n_blocks_global,
                          norm_layer)                      # incl. its head, as the reference builds (and discards) it
        self.model_downsample, self.model_resnet, self.model_upsample = g.model_downsample, g.model_resnet, g.model_upsample
        for n in range(1, n_local_enhancers + 1):
            nf = ngf * (2 ** (n_local_enhancers - n))
            if n == n_local_enhancers:
                input_nc += nz
            down = _stem7(input_nc, nf, norm_layer) + _down3(nf, nf * 2, norm_layer)
            cin = nf * 2 + (nz if n == 1 else 0)
            up = [ResnetBlock(cin, padding_type=padding_type, norm_layer=norm_layer) for _ in range(n_blocks_local)]
            up += _up3(cin, nf, norm_layer)
            setattr(self, "model%d_1" % n, nn.Sequential(*down))
            setattr(self, "model%d_2" % n, nn.Sequential(*up))
        self.model_final = nn.Sequential(*_head7(ngf + nz, output_nc, nn.Tanh()))
        self.downsample = nn.AvgPool2d(3, stride=2, padding=[1, 1], count_include_pad=False)

    def emit(self, eng, x, z, tag="Lz"):
        L = self.n_local_enhancers
        pyr = [x]
        for _ in range(L):
            pyr.append(eng.avgpool_nhwc(pyr[-1]))
        z_local = z
        for _ in range(L):
            z_local = eng.avgpool_nhwc(z_local)
        z_global = z_local
        for _ in range(self.n_downsample_global):
            z_global = eng.avgpool_nhwc(z_global)
        h = eng.run_sequential(self.model_downsample, eng.concat(pyr[-1], z_local), name=tag + ".gdown")
        h = eng.run_sequential(self.model_resnet, eng.concat(h, z_global), name=tag + ".gres")
        out = eng.run_sequential(self.model_upsample, eng.concat(h, z_global), name=tag + ".gup")
        for n in range(1, L + 1):
            inp = pyr[L - n]
            if n == L:
                inp = eng.concat(inp, z)
            comb = eng.run_sequential(getattr(self, "model%d_1" % n), inp, extra_add=out, name="%s.l%d_1" % (tag, n))
            if n == 1:
                comb = eng.concat(comb, z_local)
            out = eng.run_sequential(getattr(self, "model%d_2" % n), comb, name="%s.l%d_2" % (tag, n))
        return eng.run_sequential(self.model_final, eng.concat(out, z), head_nchw=True, name=tag + ".final")

    def forward(self, x, z):
        eng = module_engine(self, x.device)
        with torch.no_grad():
            return self.emit(eng, eng.pack(x.contiguous().float()), eng.pack(z.contiguous().float()))


class Encoder(nn.Module):
    """models/networks.py:595-632: conv encoder-decoder followed by instance-wise average pooling of its output."""

    def __init__(self, input_nc, output_nc, ngf=32, n_downsampling=4, norm_layer=nn.BatchNorm2d):
        super().__init__()
        self.output_nc = output_nc
        model = _stem7(input_nc, ngf, norm_layer)
        for i in range(n_downsampling):
            model += _down3(ngf * 2 ** i, ngf * 2 ** (i + 1), norm_layer)
        for i in range(n_downsampling):
            m = 2 ** (n_downsampling - i)
            model += _up3(ngf * m, int(ngf * m / 2), norm_layer)
        model += _head7(ngf, output_nc, nn.Tanh())
        self.model = nn.Sequential(*model)

    def forward(self, input, inst):
        """(N, output_nc, H, W) feature map, constant over every instance of `inst` (N, 1, H, W)."""
        eng = module_engine(self, input.device)
        with torch.no_grad():
            feat = eng.run_sequential(self.model, eng.pack(input.contiguous().float()), head_nchw=True, name="E")
            return eng.instance_mean(feat, inst.to(feat.device))


# --------------------------------------------------------------------------------------
# discriminators (models/networks.py:634-725)
# --------------------------------------------------------------------------------------
class NLayerDiscriminator(nn.Module):
    def __init__(self, input_nc, ndf=64, n_layers=3, norm_layer=nn.BatchNorm2d, getIntermFeat=False):
        super().__init__()
        self.getIntermFeat = getIntermFeat
        self.n_layers = n_layers
        kw = 4
        padw = int(np.ceil((kw - 1.0) / 2))
        groups = [[nn.Conv2d(input_nc, ndf, kernel_size=kw, stride=2, padding=padw), nn.LeakyReLU(0.2, True)]]
        nf = ndf
        for n in range(1, n_layers):
            nf_prev, nf = nf, min(nf * 2, 512)
            groups.append([nn.Conv2d(nf_prev, nf, kernel_size=kw, stride=2, padding=padw), norm_layer(nf),
                           nn.LeakyReLU(0.2, True)])
        nf_prev, nf = nf, min(nf * 2, 512)
        groups.append([nn.Conv2d(nf_prev, nf, kernel_size=kw, stride=1, padding=padw), norm_layer(nf),
                       nn.LeakyReLU(0.2, True)])
        groups.append([nn.Conv2d(nf, 1, kernel_size=kw, stride=1, padding=padw)])
        if getIntermFeat:
            for n, g in enumerate(groups):
                setattr(self, "model" + str(n), nn.Sequential(*g))
        else:
            self.model = nn.Sequential(*[m for g in groups for m in g])

    def groups(self):
        if self.getIntermFeat:
            return [getattr(self, "model" + str(n)) for n in range(self.n_layers + 2)]
        return [self.model]


class MultiscaleDiscriminator(nn.Module):
    def __init__(self, input_nc, ndf=64, n_layers=3, norm_layer=nn.BatchNorm2d, num_D=3, getIntermFeat=False):
        super().__init__()
        self.num_D = num_D
        self.n_layers = n_layers
        self.getIntermFeat = getIntermFeat
        ndf_max = 64
        for i in range(num_D):
            netD = NLayerDiscriminator(input_nc, min(ndf_max, ndf * (2 ** (num_D - 1 - i))), n_layers, norm_layer,
                                       getIntermFeat)
            if getIntermFeat:
                for j in range(n_layers + 2):
                    setattr(self, "scale%d_layer%d" % (i, j), getattr(netD, "model" + str(j)))
            else:
                setattr(self, "layer" + str(i), netD.model)
        self.downsample = nn.AvgPool2d(3, stride=2, padding=[1, 1], count_include_pad=False)

    def scale_groups(self, i):
        if self.getIntermFeat:
            return [getattr(self, "scale%d_layer%d" % (i, j)) for j in range(self.n_layers + 2)]
        return [getattr(self, "layer" + str(i))]

    def emit(self, eng, x, tag="D"):
        """x: Act (NHWC).  Returns list[num_D] of list of Act feature maps (last = prediction)."""
        result = []
        cur = x
        for i in range(self.num_D):
            feats = []
            h = cur
            for j, g in enumerate(self.scale_groups(self.num_D - 1 - i)):
                h = eng.run_sequential(g, h, name="%s.s%d.l%d" % (tag, self.num_D - 1 - i, j))
                feats.append(h)
            result.append(feats)
            if i != self.num_D - 1:
                cur = eng.avgpool_nhwc(cur)
        return result

    def forward(self, input):
        """NCHW fp32 in; list[num_D] of list of NCHW fp32 feature maps out (reference layout)."""
        eng = module_engine(self, input.device)
        res = self.emit(eng, eng.pack(input.contiguous().float()))
        return [[eng.unpack(f) for f in feats] for feats in res]


# --------------------------------------------------------------------------------------
# VGG19 perceptual loss (models/networks.py:776-791, 840-870)
# --------------------------------------------------------------------------------------
# torchvision.models.vgg19().features (configuration 'E' of Simonyan & Zisserman; torchvision is an external
# dependency of the reference, absent here): index -> module, 'M' = MaxPool2d(2, 2), every conv is
# Conv2d(k=3, padding=1) followed by ReLU(inplace).  The reference cuts it at indices 2 / 7 / 12 / 21 / 30.
_VGG19_CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512, "M"]
_VGG19_SLICES = [(0, 2), (2, 7), (7, 12), (12, 21), (21, 30)]


def _vgg19_feature_modules():
    mods, cin = [], 3
    for v in _VGG19_CFG:
        if v == "M":
            mods.append(nn.MaxPool2d(kernel_size=2, stride=2))
        else:
            mods += [nn.Conv2d(cin, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
            cin = v
    return mods


class Vgg19(nn.Module):
    """Frozen VGG19 feature extractor with the reference's slice layout: parameters are named
    `slice{1..5}.<features index>.{weight,bias}` exactly like models/networks.py:840-870, and `load_features`
    accepts torchvision's own `vgg19` state_dict (`features.<index>.*`).  The pretrained weights are an external
    download (torchvision model zoo) -- `VGGLoss` takes a path; random init is only for benchmarks / tests."""

    def __init__(self, requires_grad=False):
        super().__init__()
        feats = _vgg19_feature_modules()
        for k, (lo, hi) in enumerate(_VGG19_SLICES):
            seq = nn.Sequential()
            for x in range(lo, hi):
                seq.add_module(str(x), feats[x])
            setattr(self, "slice%d" % (k + 1), seq)
        for m in self.modules():                      # torchvision's VGG._initialize_weights
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
                nn.init.constant_(m.bias, 0)
        if not requires_grad:
            for p in self.parameters():
                p.requires_grad = False

    def load_features(self, state_dict):
        own = self.state_dict()
        mapped = {}
        for k, v in state_dict.items():
            if k.startswith("features."):
                idx = int(k.split(".")[1])
                for s, (lo, hi) in enumerate(_VGG19_SLICES):
                    if lo <= idx < hi:
                        mapped["slice%d.%s" % (s + 1, k[len("features."):])] = v
            elif k in own:
                mapped[k] = v
        self.load_state_dict(mapped)

    def emit(self, eng, x, tag="vgg"):
        """x: Act (NHWC, 3 channels).  Returns [h_relu1 .. h_relu5] as Acts."""
        outs = []
        for k in range(5):
            mods = list(getattr(self, "slice%d" % (k + 1)))
            i = 0
            while i < len(mods):
                m = mods[i]
                if isinstance(m, nn.MaxPool2d):
                    x = eng.maxpool2_nhwc(x)
                    i += 1
                else:                               # Conv2d + ReLU, bias and activation in the conv epilogue
                    x = eng.conv_group(x, m, L.PAD_ZERO, None, None, L.ACT_RELU, 0.0, label="%s.s%d.%d" % (tag, k + 1, i))
                    i += 2
            outs.append(x)
        return outs

    def forward(self, X):
        eng = module_engine(self, X.device)
        return [eng.unpack(f) for f in self.emit(eng, eng.pack(X.contiguous().float()))]


class VGGLoss(nn.Module):
    """sum_i w_i * L1(vgg_i(x), vgg_i(y).detach()), w = 1/32, 1/16, 1/8, 1/4, 1; both inputs are average-pooled 2x
    while wider than 1024 px (models/networks.py:776-791).  Returns a (1,1) fp32 tensor."""

    def __init__(self, gpu_id=0, checkpoint="", random_init_ok=False):
        super().__init__()
        import os
        self.vgg = Vgg19()
        if checkpoint and os.path.isfile(checkpoint):
            self.vgg.load_features(torch.load(checkpoint, map_location="cpu"))
        elif not random_init_ok:
            raise RuntimeError("VGG19 weights not found (%r): the reference downloads torchvision's pretrained vgg19; "
                               "pass --vgg19_checkpoint <vgg19-dcbb9e9d.pth> or train with --no_vgg" % checkpoint)
        if gpu_id is not None and gpu_id >= 0 and torch.cuda.is_available():
            self.vgg.cuda(gpu_id)
        self.weights = [1.0 / 32, 1.0 / 16, 1.0 / 8, 1.0 / 4, 1.0]

    def _prep(self, eng, x):
        x = x.contiguous().float()
        while x.size(3) > 1024:
            x = eng.avgpool2_planar(x)
        return x

    def features(self, x):
        """VGG features of a target image batch (no gradient): computed once per frame and shared by the
        fake_B and fake_B_raw terms (vid2vid_model_D.py:136,144 evaluate vgg(real_B) twice)."""
        eng = module_engine(self, x.device)
        with torch.no_grad():
            return self.vgg.emit(eng, eng.pack(self._prep(eng, x)), tag="vgg.y")

    def forward(self, x, y, y_feats=None):
        from . import autograd as AG
        eng = module_engine(self, x.device)
        if y_feats is None:
            y_feats = self.features(y)
        x_feats = self.vgg.emit(eng, eng.pack(self._prep(eng, x)), tag="vgg.x")
        loss = None
        for i in range(len(x_feats)):
            l = AG.l1_act(eng, x_feats[i], y_feats[i], weight=self.weights[i])
            loss = l if loss is None else loss + l
        return loss
