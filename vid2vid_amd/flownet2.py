"""FlowNet2 (forward only) on the MI355X backend.

vid2vid runs FlowNet2 frozen, eval(), under no_grad to produce the reference flow and confidence for
its losses (models/flownet.py:18-26,55).  This file re-assembles the network
(models/flownet2_pytorch/models.py:30-161 and networks/{FlowNetC,FlowNetS,FlowNetSD,FlowNetFusion,
submodules}.py) from parameter containers whose names equal the reference's -- so
`FlowNet2_checkpoint.pth.tar['state_dict']` loads by name -- and lowers it to libv2v_hip.so launches:

  * every conv / deconv (+ LeakyReLU(0.1)) is one implicit-GEMM MFMA launch with the activation fused
    in the epilogue (batchNorm=False in vid2vid's FlowNet2);
  * the two image streams of FlowNetC share weights and run as ONE batch of 2B;
  * torch.cat along channels = v2v_concat_channels_nhwc into a zero-padded NHWC buffer;
  * the ~40 pointwise ATen calls of FlowNet2.forward (mean / sub / div / cat / Upsample / Resample2d /
    ChannelNorm) are 5 fused streaming kernels (csrc/flow_ops.hip);
  * Correlation / Resample2d / ChannelNorm: v2v_correlation_forward, v2v_warp_diff_norm, v2v_channelnorm_forward.
"""
import ctypes as C

import torch
import torch.nn as nn

from . import lib as L
from .lib import lib, check
from .engine import Act, pad_channels, _ptr, _stream


# --------------------------------------------------------------------------------------
# parameter containers (names == reference state_dict keys)
# --------------------------------------------------------------------------------------
def conv(cin, cout, kernel_size=3, stride=1):
    """submodules.py:7-19 with batchNorm=False"""
    return nn.Sequential(nn.Conv2d(cin, cout, kernel_size, stride=stride, padding=(kernel_size - 1) // 2, bias=True),
                         nn.LeakyReLU(0.1, inplace=True))


def i_conv(cin, cout):
    """submodules.py:21-31 with batchNorm=False"""
    return nn.Sequential(nn.Conv2d(cin, cout, 3, stride=1, padding=1, bias=True))


def predict_flow(cin):
    return nn.Conv2d(cin, 2, kernel_size=3, stride=1, padding=1, bias=True)


def deconv(cin, cout):
    return nn.Sequential(nn.ConvTranspose2d(cin, cout, kernel_size=4, stride=2, padding=1, bias=True),
                         nn.LeakyReLU(0.1, inplace=True))


def _decoder_params(m, up_bias, with_inter):
    m.deconv5, m.deconv4, m.deconv3, m.deconv2 = deconv(1024, 512), deconv(1026, 256), deconv(770, 128), deconv(386, 64)
    if with_inter:
        m.inter_conv5, m.inter_conv4 = i_conv(1026, 512), i_conv(770, 256)
        m.inter_conv3, m.inter_conv2 = i_conv(386, 128), i_conv(194, 64)
        chans = (1024, 512, 256, 128, 64)
    else:
        chans = (1024, 1026, 770, 386, 194)
    for lvl, c in zip((6, 5, 4, 3, 2), chans):
        setattr(m, "predict_flow%d" % lvl, predict_flow(c))
    for a, b in ((6, 5), (5, 4), (4, 3), (3, 2)):
        setattr(m, "upsampled_flow%d_to_%d" % (a, b), nn.ConvTranspose2d(2, 2, 4, 2, 1, bias=up_bias))


class FlowNetC(nn.Module):
    """networks/FlowNetC.py:13-131"""

    def __init__(self):
        super().__init__()
        self.conv1, self.conv2, self.conv3 = conv(3, 64, 7, 2), conv(64, 128, 5, 2), conv(128, 256, 5, 2)
        self.conv_redir = conv(256, 32, 1, 1)
        self.conv3_1 = conv(473, 256)
        self.conv4, self.conv4_1 = conv(256, 512, stride=2), conv(512, 512)
        self.conv5, self.conv5_1 = conv(512, 512, stride=2), conv(512, 512)
        self.conv6, self.conv6_1 = conv(512, 1024, stride=2), conv(1024, 1024)
        _decoder_params(self, up_bias=True, with_inter=False)


class FlowNetS(nn.Module):
    """networks/FlowNetS.py:15-93"""

    def __init__(self, input_channels=12):
        super().__init__()
        self.conv1, self.conv2 = conv(input_channels, 64, 7, 2), conv(64, 128, 5, 2)
        self.conv3, self.conv3_1 = conv(128, 256, 5, 2), conv(256, 256)
        self.conv4, self.conv4_1 = conv(256, 512, stride=2), conv(512, 512)
        self.conv5, self.conv5_1 = conv(512, 512, stride=2), conv(512, 512)
        self.conv6, self.conv6_1 = conv(512, 1024, stride=2), conv(1024, 1024)
        _decoder_params(self, up_bias=False, with_inter=False)


class FlowNetSD(nn.Module):
    """networks/FlowNetSD.py:11-107"""

    def __init__(self):
        super().__init__()
        self.conv0 = conv(6, 64)
        self.conv1, self.conv1_1 = conv(64, 64, stride=2), conv(64, 128)
        self.conv2, self.conv2_1 = conv(128, 128, stride=2), conv(128, 128)
        self.conv3, self.conv3_1 = conv(128, 256, stride=2), conv(256, 256)
        self.conv4, self.conv4_1 = conv(256, 512, stride=2), conv(512, 512)
        self.conv5, self.conv5_1 = conv(512, 512, stride=2), conv(512, 512)
        self.conv6, self.conv6_1 = conv(512, 1024, stride=2), conv(1024, 1024)
        _decoder_params(self, up_bias=True, with_inter=True)


class FlowNetFusion(nn.Module):
    """networks/FlowNetFusion.py:11-69"""

    def __init__(self):
        super().__init__()
        self.conv0 = conv(11, 64)
        self.conv1, self.conv1_1 = conv(64, 64, stride=2), conv(64, 128)
        self.conv2, self.conv2_1 = conv(128, 128, stride=2), conv(128, 128)
        self.deconv1, self.deconv0 = deconv(128, 32), deconv(162, 16)
        self.inter_conv1, self.inter_conv0 = i_conv(162, 32), i_conv(82, 16)
        self.predict_flow2, self.predict_flow1, self.predict_flow0 = predict_flow(128), predict_flow(32), predict_flow(16)
        self.upsampled_flow2_to_1 = nn.ConvTranspose2d(2, 2, 4, 2, 1)
        self.upsampled_flow1_to_0 = nn.ConvTranspose2d(2, 2, 4, 2, 1)


class FlowNet2(nn.Module):
    """models/flownet2_pytorch/models.py:30-161 (batchNorm=False, div_flow=20, rgb_max=1 as vid2vid builds it,
    models/flownet.py:18)."""

    def __init__(self, div_flow=20.0, rgb_max=1.0):
        super().__init__()
        self.div_flow, self.rgb_max = div_flow, rgb_max
        self.flownetc = FlowNetC()
        self.flownets_1 = FlowNetS()
        self.flownets_2 = FlowNetS()
        self.flownets_d = FlowNetSD()
        self.flownetfusion = FlowNetFusion()
        for m in self.modules():            # models.py:68-77
            if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
                if m.bias is not None:
                    nn.init.uniform_(m.bias)
                nn.init.xavier_uniform_(m.weight)

    # ------------------------------------------------------------------ lowering
    def emit(self, eng, im1, im2):
        """im1, im2: planar fp32 [B,3,H,W] (H, W multiples of 64).  Returns planar fp32 flow [B,2,H,W]."""
        R = _Runner(eng)
        B, _, H, W = im1.shape
        dv = self.div_flow
        x1, x2 = R.normalize(im1, im2, self.rgb_max)
        # ---- FlowNetSD on x (models.py:143-150): depends on the images only -- recorded on its own plan lane (round 6), it runs beside
        #      the FlowNetC -> FlowNetS -> FlowNetS chain, whose ~150 small dependent launches leave most of the chip idle ----
        xsd = R.pack_cat(B, H, W, [(x1, 1.0), (x2, 1.0)])
        with eng.on_lane(R.LANE_SD):
            flownetsd_flow = R.up4(R.unpack(self._flownetsd(R, xsd)), False, 1.0 / dv)
            norm_sd = R.channelnorm(flownetsd_flow)
            _, diff_sd = R.warp_diff(x1, x2, flownetsd_flow, want_warped=False)
        # ---- FlowNetC on the two streams (models.py:105-106) ----
        flow = R.up4(R.unpack(self._flownetc(R, x1, x2)), True, dv)
        # ---- FlowNetS 1 and 2 on [x, warped img1, flow/div, |diff|] (models.py:108-131) ----
        for net, bilinear in ((self.flownets_1, True), (self.flownets_2, False)):
            warped, ndiff = R.warp_diff(x1, x2, flow, want_warped=True)
            cat = R.pack_cat(B, H, W, [(x1, 1.0), (x2, 1.0), (warped, 1.0), (flow, 1.0 / dv), (ndiff, 1.0)])
            flow = R.up4(R.unpack(self._flownets(R, net, cat)), bilinear, dv)
        flownets2_flow = flow
        norm_s2 = R.channelnorm(flownets2_flow)
        _, diff_s2 = R.warp_diff(x1, x2, flownets2_flow, want_warped=False)
        # ---- fusion (models.py:155-156) ----
        eng.join(R.LANE_SD)
        cat3 = R.pack_cat(B, H, W, [(x1, 1.0), (flownetsd_flow, 1.0), (flownets2_flow, 1.0), (norm_sd, 1.0),
                                    (norm_s2, 1.0), (diff_sd, 1.0), (diff_s2, 1.0)])
        return R.unpack(self._fusion(R, cat3))

    @staticmethod
    def _decode(R, m, c6, skips, with_inter):
        """Shared refinement ladder: predict_flow / upsampled_flow / deconv / concat, levels 6 -> 2."""
        feat = c6
        flow = R.cv(getattr(m, "predict_flow6"), c6)
        for lvl, skip in zip((5, 4, 3, 2), skips):
            feat = R.cat_level(skip, getattr(m, "deconv%d" % lvl), feat, getattr(m, "upsampled_flow%d_to_%d" % (lvl + 1, lvl)), flow)
            head_in = R.cv(getattr(m, "inter_conv%d" % lvl), feat) if with_inter else feat
            flow = R.cv(getattr(m, "predict_flow%d" % lvl), head_in)
        return flow

    def _flownetc(self, R, x1, x2):
        m, eng = self.flownetc, R.eng
        B, _, H, W = x1.shape
        both = R.pack_batch([x1, x2])                              # [2B,H,W,3]: both streams in one launch
        c1 = R.cv(m.conv1, both)
        c2 = R.cv(m.conv2, c1)
        c3 = R.cv(m.conv3, c2)
        c2a, c3a = Act(c2.t[:B], c2.C), Act(c3.t[:B], c3.C)
        redir = R.cv(m.conv_redir, c3a)
        if R.corr_mma_ok(c3):                                      # matrix-pipe correlation straight into the concat buffer (round 3)
            in31, off = R.cat([redir, None], reserve=441)
            R.correlation_into(c3, in31, off, 0.1)                 # cat(conv_redir, leaky_relu(corr, 0.1)), FlowNetC.py:86-93
        else:
            corr = R.correlation(c3)                               # planar [B,441,h,w]
            in31 = R.cat([redir, None], extra=(corr, 441, 0.1))
        c3_1 = R.cv(m.conv3_1, in31)
        c4 = R.cv(m.conv4_1, R.cv(m.conv4, c3_1))
        c5 = R.cv(m.conv5_1, R.cv(m.conv5, c4))
        c6 = R.cv(m.conv6_1, R.cv(m.conv6, c5))
        return self._decode(R, m, c6, (c5, c4, c3_1, c2a), False)

    def _flownets(self, R, m, x):
        c1 = R.cv(m.conv1, x)
        c2 = R.cv(m.conv2, c1)
        c3 = R.cv(m.conv3_1, R.cv(m.conv3, c2))
        c4 = R.cv(m.conv4_1, R.cv(m.conv4, c3))
        c5 = R.cv(m.conv5_1, R.cv(m.conv5, c4))
        c6 = R.cv(m.conv6_1, R.cv(m.conv6, c5))
        return self._decode(R, m, c6, (c5, c4, c3, c2), False)

    def _flownetsd(self, R, x):
        m = self.flownets_d
        c0 = R.cv(m.conv0, x)
        c1 = R.cv(m.conv1_1, R.cv(m.conv1, c0))
        c2 = R.cv(m.conv2_1, R.cv(m.conv2, c1))
        c3 = R.cv(m.conv3_1, R.cv(m.conv3, c2))
        c4 = R.cv(m.conv4_1, R.cv(m.conv4, c3))
        c5 = R.cv(m.conv5_1, R.cv(m.conv5, c4))
        c6 = R.cv(m.conv6_1, R.cv(m.conv6, c5))
        return self._decode(R, m, c6, (c5, c4, c3, c2), True)

    def _fusion(self, R, x):
        m = self.flownetfusion
        c0 = R.cv(m.conv0, x)
        c1 = R.cv(m.conv1_1, R.cv(m.conv1, c0))
        c2 = R.cv(m.conv2_1, R.cv(m.conv2, c1))
        flow2 = R.cv(m.predict_flow2, c2)
        cat1 = R.cat_level(c1, m.deconv1, c2, m.upsampled_flow2_to_1, flow2)
        flow1 = R.cv(m.predict_flow1, R.cv(m.inter_conv1, cat1))
        cat0 = R.cat_level(c0, m.deconv0, cat1, m.upsampled_flow1_to_0, flow1)
        return R.cv(m.predict_flow0, R.cv(m.inter_conv0, cat0))


class _Runner:
    """Launch emitters for the FlowNet2 glue (all recordable into a Plan)."""

    def __init__(self, eng):
        import os
        self.eng = eng
        self.direct_cat = os.environ.get("V2V_FLOWNET_DIRECT_CAT", "1") != "0"
        self.fork_decoder = os.environ.get("V2V_FLOWNET_FORK", "0") == "1"

    def _f32(self, *shape):
        return self.eng.empty_f32(*shape)

    LANE_SD = 1          # FlowNetSD; the refinement ladders fork their deconv branch onto lane (current + 2)

    def cv(self, mod, x, out=None):
        """conv / deconv (+ LeakyReLU(0.1) when the container has one), activation fused in the epilogue.  out: an Act view
        (channel range of a concat buffer) the launch writes instead of a tensor of its own."""
        if isinstance(mod, nn.Sequential):
            cmod = mod[0]
            act = (L.ACT_LEAKY, 0.1) if len(mod) > 1 else (L.ACT_NONE, 0.0)
        else:
            cmod, act = mod, (L.ACT_NONE, 0.0)
        if out is None:
            return self.eng.conv_group(x, cmod, L.PAD_ZERO, None, None, act[0], act[1], label="flownet")
        res, _, _ = self.eng.conv(x, cmod, L.PAD_ZERO, None, L.OUT_ACT_NHWC, act[0], act[1], 1.0, out=out, label="flownet")
        return res

    def cat_level(self, skip, deconv_mod, feat, up_mod, flow):
        """One level of a refinement ladder: cat(skip, deconv(feat), upsampled_flow(flow)) (FlowNetS.py:71-90 and the same lines
        of FlowNetC / FlowNetSD / FlowNetFusion).  Round 6: the deconvolution and the flow up-sampling WRITE their channel ranges
        of the concat buffer (whole 16-byte vectors: every offset is a multiple of 8 channels) instead of tensors of their own that
        two copy launches then moved; only the skip tensor -- which the encoder's next layer also reads, at its own stride -- is
        copied (3 pairs at 512x256: 208 -> 172 launches, 3.40 -> 3.27 ms per pass).  The deconvolution (and that copy) do not depend
        on the flow head; recorded on a forked plan lane (V2V_FLOWNET_FORK=1) the 18 fork / join edge pairs cost more than the
        overlap returns (2.93 -> 3.21 ms with FlowNetSD on its lane, profiles/r06_v36_flownet2_ab.txt): off."""
        eng = self.eng
        c_dec, c_up = deconv_mod[0].out_channels, up_mod.out_channels
        N, H, W = skip.N, skip.H, skip.W
        total = skip.C + c_dec + c_up
        vec = 8 if eng.dtype == L.BF16 else 4
        if skip.C % vec != 0 or (skip.C + c_dec) % vec != 0 or not self.direct_cat:
            return self.cat([skip, self.cv(deconv_mod, feat), self.cv(up_mod, flow)])
        out = self._zeros_act(N, H, W, total)
        side = eng._lane + 2 if self.fork_decoder else eng._lane
        with eng.on_lane(side):
            check(lib.v2v_concat_channels_nhwc(_ptr(skip.t), skip.Cs, 0, _ptr(out.t), out.Cs, 0, skip.C, N * H * W,
                                               eng.dtype, _stream()), "concat_channels")
            eng.label("concat_channels_nhwc")
            self.cv(deconv_mod, feat, out=Act(out.t[..., skip.C:skip.C + c_dec], c_dec))
        self.cv(up_mod, flow, out=Act(out.t[..., skip.C + c_dec:], c_up))
        eng.join(side)
        return out

    def normalize(self, im1, im2, rgb_max):
        B, _, H, W = im1.shape
        x1, x2 = self._f32(B, 3, H, W), self._f32(B, 3, H, W)
        ws = self.eng.scratch("flownet_mean", B * 3 * 64)
        check(lib.v2v_flownet_normalize(_ptr(im1), _ptr(im2), _ptr(x1), _ptr(x2), _ptr(ws), B, H, W, float(rgb_max),
                                        _stream()), "flownet_normalize")
        self.eng.label("flownet_normalize")
        return x1, x2

    def _pack_into(self, planar, buf, n0, c_off, scale=1.0, slope=1.0):
        B, Cc, H, W = planar.shape
        dst = buf[n0:n0 + B]
        check(lib.v2v_pack_channels_nhwc(_ptr(planar), _ptr(dst), B, Cc, H, W, buf.stride(2), c_off, float(scale),
                                         float(slope), self.eng.dtype, _stream()), "pack_channels")
        self.eng.label("pack_channels_nhwc")

    def _zeros_act(self, N, H, W, Cc):
        eng = self.eng
        t = torch.zeros((N, H, W, pad_channels(Cc, eng.dtype)), dtype=eng.tdtype, device=eng.device)
        eng._keep(t)
        return Act(t, Cc)

    def pack_cat(self, B, H, W, parts):
        """cat of planar tensors (each times a scale) -> NHWC Act"""
        total = sum(p.shape[1] for p, _ in parts)
        out = self._zeros_act(B, H, W, total)
        off = 0
        for p, scale in parts:
            self._pack_into(p, out.t, 0, off, scale)
            off += p.shape[1]
        return out

    def pack_batch(self, planars):
        """stack planar tensors along the batch dim -> NHWC Act [len*B,H,W,C]"""
        B, Cc, H, W = planars[0].shape
        out = self._zeros_act(B * len(planars), H, W, Cc)
        for i, p in enumerate(planars):
            self._pack_into(p, out.t, i * B, 0)
        return out

    def cat(self, acts, extra=None, reserve=0):
        """torch.cat along channels of NHWC Acts; `extra` = (planar tensor, C, leaky slope) appended last
        (replaces a None placeholder); `reserve` = channels left for a producer that writes into the buffer itself
        (returns (Act, channel offset of the reserved range))."""
        eng = self.eng
        parts = [a for a in acts if a is not None]
        N, H, W = parts[0].N, parts[0].H, parts[0].W
        total = sum(a.C for a in parts) + (extra[1] if extra else 0) + reserve
        out = self._zeros_act(N, H, W, total)
        off = 0
        for a in parts:
            check(lib.v2v_concat_channels_nhwc(_ptr(a.t), a.Cs, 0, _ptr(out.t), out.Cs, off, a.C, N * H * W,
                                               eng.dtype, _stream()), "concat_channels")
            eng.label("concat_channels_nhwc")
            off += a.C
        if extra:
            self._pack_into(extra[0], out.t, 0, off, 1.0, extra[2])
        if reserve:
            return out, off
        return out

    def unpack(self, a):
        return self.eng.unpack(a)

    def up4(self, planar, bilinear, scale):
        """nn.Upsample(scale_factor=4) * scale (models.py:49-61)"""
        B, Cc, h, w = planar.shape
        out = self._f32(B, Cc, 4 * h, 4 * w)
        check(lib.v2v_resize_planar(_ptr(planar), _ptr(out), B * Cc, h, w, 4 * h, 4 * w, int(bilinear), float(scale),
                                    _stream()), "resize_planar")
        self.eng.label("resize_planar")
        return out

    def warp_diff(self, x1, x2, flow, want_warped):
        B, _, H, W = x1.shape
        warped = self._f32(B, 3, H, W) if want_warped else None
        nrm = self._f32(B, 1, H, W)
        check(lib.v2v_warp_diff_norm(_ptr(x1), 3 * H * W, _ptr(x2), 3 * H * W, _ptr(flow), _ptr(warped), _ptr(nrm),
                                     B, 3, H, W, 0, 0.0, _stream()), "warp_diff_norm")
        self.eng.label("warp_diff_norm")
        return warped, nrm

    def channelnorm(self, x):
        B, Cc, H, W = x.shape
        out = self._f32(B, 1, H, W)
        check(lib.v2v_channelnorm_forward(_ptr(x), _ptr(out), B, Cc, H, W, 2, _stream()), "channelnorm")
        self.eng.label("channelnorm")
        return out

    def corr_mma_ok(self, c3):
        """v2v_correlation_nhwc: channels a whole number of MFMA K steps (FlowNetC: 256).  V2V_CORR_MMA=0: the planar C-ABI op."""
        import os
        ks = 16 if self.eng.dtype == L.BF16 else 8
        return os.environ.get("V2V_CORR_MMA", "1") != "0" and c3.C % ks == 0 and c3.N % 2 == 0

    def correlation_into(self, c3, out, c_off, slope):
        """Correlation(pad 20, k 1, max_disp 20, stride1 1, stride2 2) of the two halves of the stacked conv3 output + LeakyReLU
        (FlowNetC.py:31,86-89), written as 441 channels at `c_off` of the NHWC concat buffer `out`."""
        eng = self.eng
        B = c3.N // 2
        check(lib.v2v_correlation_nhwc(_ptr(c3.t[:B]), _ptr(c3.t[B:]), _ptr(out.t), B, c3.C, c3.H, c3.W, c3.Cs, out.Cs, c_off,
                                       20, 2, float(slope), eng.dtype, _stream()), "correlation_nhwc")
        eng.label("correlation")

    def correlation(self, c3):
        """Correlation(pad 20, k 1, max_disp 20, stride1 1, stride2 2) between the two halves of the
        stacked conv3 output (FlowNetC.py:31,86-88)."""
        eng = self.eng
        planar = eng.unpack(c3)                  # [2B,256,h,w]
        B2, Cc, h, w = planar.shape
        B = B2 // 2
        oc, oh, ow = C.c_int32(), C.c_int32(), C.c_int32()
        lib.v2v_correlation_out_size(h, w, 20, 1, 20, 1, 2, C.byref(oc), C.byref(oh), C.byref(ow))
        out = self._f32(B, oc.value, oh.value, ow.value)
        check(lib.v2v_correlation_forward(_ptr(planar[:B]), _ptr(planar[B:]), _ptr(out), B, Cc, h, w, 20, 1, 20, 1, 2, 1,
                                          _stream()), "correlation")
        eng.label("correlation")
        return out
