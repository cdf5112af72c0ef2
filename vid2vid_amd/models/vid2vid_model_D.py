"""Vid2VidModelD on the MI355X backend.

API-compatible with the reference's models/vid2vid_model_D.py: `initialize(opt)`,
`forward(scale_T, tensors_list) -> [loss (1,1) tensors]` in the order of `loss_names` /
`loss_names_T`, `get_all_skipped_frames`, `get_losses`, `save`, optimizers `optimizer_D`,
`optimizer_D_T{s}`.  Every discriminator forward, loss reduction and their backward passes are
libv2v_hip.so launches recorded as `v2v` custom ops (autograd.py); the only torch arithmetic left is
the addition / scaling of the (1,1) loss scalars, exactly where the reference has it.
"""
import torch

from .. import autograd as AG
from .. import networks
from ..optim import FusedAdam
from .base_model import BaseModel


class Vid2VidModelD(BaseModel):
    def name(self):
        return "Vid2VidModelD"

    def initialize(self, opt):
        BaseModel.initialize(self, opt)
        self.tD = opt.n_frames_D
        self.output_nc = opt.output_nc
        self.n_scales = opt.n_scales_spatial
        if opt.add_face_disc:
            raise NotImplementedError("--add_face_disc (pose recipes) is outside the MI355X hot path (SURVEY 8a13)")
        if not opt.no_vgg:          # reference :66-67; the weights are torchvision's download -> a path option here
            self.criterionVGG = networks.VGGLoss(self.device.index if self.device.type == "cuda" else -1,
                                                 getattr(opt, "vgg19_checkpoint", "checkpoints/vgg19-dcbb9e9d.pth"),
                                                 getattr(opt, "random_init_ok", False))
        if opt.gan_mode != "ls":
            raise NotImplementedError("only the LSGAN objective (--gan_mode ls, the reference default) is implemented")

        # single-image discriminator (reference :29-36)
        self.input_nc = opt.label_nc if opt.label_nc != 0 else opt.input_nc
        if opt.use_instance:
            self.input_nc += 1
        netD_input_nc = self.input_nc + opt.output_nc
        gpu_ids = self.dev_ids
        self.netD = networks.define_D(netD_input_nc, opt.ndf, opt.n_layers_D, opt.norm, opt.num_D,
                                      not opt.no_ganFeat, gpu_ids=gpu_ids)
        # temporal discriminators (reference :42-46)
        netD_input_nc = opt.output_nc * opt.n_frames_D + 2 * (opt.n_frames_D - 1)
        for s in range(opt.n_scales_temporal):
            setattr(self, "netD_T" + str(s),
                    networks.define_D(netD_input_nc, opt.ndf, opt.n_layers_D, opt.norm, opt.num_D,
                                      not opt.no_ganFeat, gpu_ids=gpu_ids))
        print("---------- Networks initialized -------------")

        if opt.continue_train or opt.load_pretrain:
            self.load_network(self.netD, "D", opt.which_epoch, opt.load_pretrain)
            for s in range(opt.n_scales_temporal):
                self.load_network(getattr(self, "netD_T" + str(s)), "D_T" + str(s), opt.which_epoch, opt.load_pretrain)

        self.old_lr = opt.lr
        self.loss_names = ["G_VGG", "G_GAN", "G_GAN_Feat", "D_real", "D_fake", "G_Warp", "F_Flow", "F_Warp", "W"]
        self.loss_names_T = ["G_T_GAN", "G_T_GAN_Feat", "D_T_real", "D_T_fake", "G_T_Warp"]

        if opt.TTUR:
            beta1, beta2, lr = 0, 0.9, opt.lr * 2
        else:
            beta1, beta2, lr = opt.beta1, 0.999, opt.lr
        self.optimizer_D = FusedAdam(list(self.netD.parameters()), lr=lr, betas=(beta1, beta2))
        for s in range(opt.n_scales_temporal):
            params = list(getattr(self, "netD_T" + str(s)).parameters())
            setattr(self, "optimizer_D_T" + str(s), FusedAdam(params, lr=opt.lr, betas=(opt.beta1, 0.999)))
        self.bind_precision()

    # ------------------------------------------------------------------ losses
    def _zero(self):
        return torch.zeros(1, 1, dtype=torch.float32, device=self.device)

    def criterionGAN(self, preds, target_is_real):
        """GANLoss.__call__ for multiscale predictions: sum over scales of MSE(pred_last, 1 or 0)
        (models/networks.py:764-771)."""
        eng = self.engine
        target = 1.0 if target_is_real else 0.0
        loss = None
        for feats in preds:
            l = AG.mse_const_act(eng, feats[-1], target)
            loss = l if loss is None else loss + l
        return loss

    def GAN_and_FM_loss(self, pred_real, pred_fake):
        """reference :199-213"""
        opt, eng = self.opt, self.engine
        loss_G_GAN = self.criterionGAN(pred_fake, True)
        loss_FM = self._zero()
        if not opt.no_ganFeat:
            w = (4.0 / (opt.n_layers_D + 1)) * (1.0 / opt.num_D) * opt.lambda_feat
            for i in range(min(len(pred_fake), opt.num_D)):
                for j in range(len(pred_fake[i]) - 1):
                    loss_FM = loss_FM + AG.l1_act(eng, pred_fake[i][j], pred_real[i][j], weight=w)
        return loss_G_GAN, loss_FM

    def _run_D(self, netD, x0, x1, scale1=1.0, tag="D"):
        eng = self.engine
        return netD.emit(eng, AG.pack_concat(eng, x0, x1, scale1), tag=tag)

    def compute_loss_D(self, netD, real_A, real_B, fake_B):
        """reference :168-179 -- three discriminator passes (real, fake detached, fake)."""
        pred_real = self._run_D(netD, real_A, real_B)
        pred_fake = self._run_D(netD, real_A, fake_B.detach())
        loss_D_real = self.criterionGAN(pred_real, True)
        loss_D_fake = self.criterionGAN(pred_fake, False)
        pred_fake = self._run_D(netD, real_A, fake_B)
        loss_G_GAN, loss_G_GAN_Feat = self.GAN_and_FM_loss(pred_real, pred_fake)
        return loss_D_real, loss_D_fake, loss_G_GAN, loss_G_GAN_Feat

    def compute_loss_D_T(self, real_B, fake_B, flow_ref, conf_ref, scale_T, flow_scale):
        """reference :181-197; flow_ref arrives unscaled, the /20 of :107-108 is folded into the pack."""
        netD_T = getattr(self, "netD_T" + str(scale_T))
        H, W = self.height, self.width
        real_B = real_B.reshape(-1, self.output_nc * self.tD, H, W)
        fake_B = fake_B.reshape(-1, self.output_nc * self.tD, H, W)
        if flow_ref is not None:
            flow_ref = flow_ref.reshape(-1, 2 * (self.tD - 1), H, W)
        tag = "D_T%d" % scale_T
        pred_real = self._run_D(netD_T, real_B, flow_ref, flow_scale, tag)
        pred_fake = self._run_D(netD_T, fake_B.detach(), flow_ref, flow_scale, tag)
        loss_D_T_real = self.criterionGAN(pred_real, True)
        loss_D_T_fake = self.criterionGAN(pred_fake, False)
        pred_fake = self._run_D(netD_T, fake_B, flow_ref, flow_scale, tag)
        loss_G_T_GAN, loss_G_T_GAN_Feat = self.GAN_and_FM_loss(pred_real, pred_fake)
        return loss_D_T_real, loss_D_T_fake, loss_G_T_GAN, loss_G_T_GAN_Feat

    def forward(self, scale_T, tensors_list, dummy_bs=0):
        opt, eng = self.opt, self.engine
        lambda_feat, lambda_F, lambda_T = opt.lambda_feat, opt.lambda_F, opt.lambda_T
        scale_S = opt.n_scales_spatial
        if dummy_bs:
            tensors_list = [None if t is None else t[dummy_bs:] for t in tensors_list]
        dev = self.device

        def dv(t):
            return None if t is None else t.to(dev, torch.float32)

        if scale_T > 0:
            real_B, fake_B, flow_ref, conf_ref = [dv(t) for t in tensors_list]
            _, _, _, self.height, self.width = real_B.size()
            l_real, l_fake, l_gan, l_feat = self.compute_loss_D_T(real_B, fake_B, flow_ref, conf_ref, scale_T - 1, 1.0 / 20.0)
            loss_list = [l_gan, l_feat, l_real, l_fake, self._zero()]
            return [l.view(-1, 1) for l in loss_list]

        (real_B, fake_B, fake_B_raw, real_A, real_B_prev, fake_B_prev, flow, weight, flow_ref, conf_ref) = \
            [dv(t) for t in tensors_list]
        _, _, self.height, self.width = real_B.size()

        # ---- flow losses (reference :114-131) ----
        if flow is not None:
            loss_F_Flow = AG.masked_l1(eng, flow, flow_ref, conf_ref, weight=lambda_F / (2 ** (scale_S - 1)))
            real_B_warp = eng.resample_flow(real_B_prev.contiguous(), flow.contiguous())
            loss_F_Warp = AG.masked_l1(eng, real_B_warp, real_B, conf_ref, weight=lambda_T)
            loss_W = self._zero()
            if opt.no_first_img:
                loss_W = AG.masked_l1(eng, weight, torch.zeros_like(weight), conf_ref)
        else:
            loss_F_Flow, loss_F_Warp, loss_W = self._zero(), self._zero(), self._zero()

        # ---- image GAN + feature matching (reference :133-147) ----
        loss_G_VGG = self._zero()
        vgg_real = None
        if not opt.no_vgg:
            vgg_real = self.criterionVGG.features(real_B)            # shared by the fake_B and fake_B_raw terms
            loss_G_VGG = self.criterionVGG(fake_B, real_B, vgg_real) * lambda_feat
        loss_D_real, loss_D_fake, loss_G_GAN, loss_G_GAN_Feat = self.compute_loss_D(self.netD, real_A, real_B, fake_B)
        with torch.no_grad():
            fake_B_warp_ref = eng.resample_flow(fake_B_prev.detach().contiguous(), flow_ref.contiguous())
        loss_G_Warp = AG.masked_l1(eng, fake_B, fake_B_warp_ref, conf_ref, weight=lambda_T)
        if fake_B_raw is not None:
            l_D_real, l_D_fake, l_G_GAN, l_G_GAN_Feat = self.compute_loss_D(self.netD, real_A, real_B, fake_B_raw)
            loss_G_GAN = loss_G_GAN + l_G_GAN
            loss_G_GAN_Feat = loss_G_GAN_Feat + l_G_GAN_Feat
            loss_D_real = loss_D_real + l_D_real
            loss_D_fake = loss_D_fake + l_D_fake
            if not opt.no_vgg:
                loss_G_VGG = loss_G_VGG + self.criterionVGG(fake_B_raw, real_B, vgg_real) * lambda_feat

        loss_list = [loss_G_VGG, loss_G_GAN, loss_G_GAN_Feat, loss_D_real, loss_D_fake,
                     loss_G_Warp, loss_F_Flow, loss_F_Warp, loss_W]
        return [l.view(-1, 1) for l in loss_list]

    # ------------------------------------------------------------------ bookkeeping used by train.py
    def get_all_skipped_frames(self, frames_all, real_B, fake_B, flow_ref, conf_ref, t_scales, tD, n_frames_load, i, flowNet):
        """Rolling frame history -> tD-strided groups per temporal scale (reference :232-247)."""
        real_B_all, fake_B_all, flow_ref_all, conf_ref_all = frames_all
        real_B_sk = fake_B_sk = flow_ref_sk = conf_ref_sk = None
        if t_scales > 0:
            if self.opt.sparse_D:
                real_B_all, real_B_sk = get_skipped_frames_sparse(real_B_all, real_B, t_scales, tD, n_frames_load, i)
                fake_B_all, fake_B_sk = get_skipped_frames_sparse(fake_B_all, fake_B, t_scales, tD, n_frames_load, i)
                flow_ref_all, flow_ref_sk = get_skipped_frames_sparse(flow_ref_all, flow_ref, t_scales, tD, n_frames_load, i, is_flow=True)
                conf_ref_all, conf_ref_sk = get_skipped_frames_sparse(conf_ref_all, conf_ref, t_scales, tD, n_frames_load, i, is_flow=True)
            else:
                real_B_all, real_B_sk = get_skipped_frames(real_B_all, real_B, t_scales, tD)
                fake_B_all, fake_B_sk = get_skipped_frames(fake_B_all, fake_B, t_scales, tD)
                flow_ref_all, conf_ref_all, flow_ref_sk, conf_ref_sk = get_skipped_flows(
                    flowNet, flow_ref_all, conf_ref_all, real_B_sk, flow_ref, conf_ref, t_scales, tD)
        return (real_B_all, fake_B_all, flow_ref_all, conf_ref_all), (real_B_sk, fake_B_sk, flow_ref_sk, conf_ref_sk)

    def get_losses(self, loss_dict, loss_dict_T, t_scales):
        """reference :249-264"""
        loss_D = (loss_dict["D_fake"] + loss_dict["D_real"]) * 0.5
        loss_G = loss_dict["G_GAN"] + loss_dict["G_GAN_Feat"] + loss_dict["G_VGG"]
        loss_G = loss_G + loss_dict["G_Warp"] + loss_dict["F_Flow"] + loss_dict["F_Warp"] + loss_dict["W"]
        loss_D_T = []
        t_scales_act = min(t_scales, len(loss_dict_T))
        for s in range(t_scales_act):
            loss_G = loss_G + loss_dict_T[s]["G_T_GAN"] + loss_dict_T[s]["G_T_GAN_Feat"] + loss_dict_T[s]["G_T_Warp"]
            loss_D_T.append((loss_dict_T[s]["D_T_fake"] + loss_dict_T[s]["D_T_real"]) * 0.5)
        return loss_G, loss_D, loss_D_T, t_scales_act

    def save(self, label):
        self.save_network(self.netD, "D", label, self.gpu_ids)
        for s in range(self.opt.n_scales_temporal):
            self.save_network(getattr(self, "netD_T" + str(s)), "D_T" + str(s), label, self.gpu_ids)


# --------------------------------------------------------------------------------------
# temporal sub-sampling of the frame history (reference :274-328): pure indexing
# --------------------------------------------------------------------------------------
def get_skipped_frames(B_all, B, t_scales, tD):
    """Append B to the history; for temporal scale s return the groups of tD frames spaced tD**s apart
    that end at each of the newest frames (stacked along dim 0)."""
    B_all = B if B_all is None else torch.cat([B_all.detach(), B], dim=1)
    skipped = [None] * t_scales
    total, new = B_all.size(1), B.size(1)
    for s in range(t_scales):
        step = tD ** s
        span = step * (tD - 1)
        n_groups = min(total - span, new)
        groups = []
        for t in range(0, max(n_groups, 0), tD):
            end = total - t                       # one past the last frame of this group
            groups.append(B_all[:, end - span - 1:end:step].contiguous())
        if groups:
            skipped[s] = groups[0] if len(groups) == 1 else torch.cat(groups)
    keep = tD ** (t_scales - 1) * (tD - 1)
    if total > keep:
        B_all = B_all[:, -keep:]
    return B_all, skipped


def get_skipped_flows(flowNet, flow_ref_all, conf_ref_all, real_B, flow_ref, conf_ref, t_scales, tD):
    """Reference flows / confidences for every temporal scale (behaviour of the reference's :292-302, written from its
    specification: temporal scale 0 looks at tD consecutive frames and re-uses the tD-1 consecutive-frame flows FlowNet2 already
    produced for the chunk; scale s >= 1 looks at tD frames that are tD**s apart, for which no flow exists yet, so
    FlowNet2 runs on those strided neighbours -- but only once a full group of tD frames is available)."""
    def scale0_groups(history, new):
        # the same grouping as the frames of scale 0; a group of tD entries holds tD - 1 flows between its members
        history, grouped = get_skipped_frames(history, new, 1, tD)
        return history, (None if grouped[0] is None else grouped[0][:, 1:])

    flow_ref_all, flow0 = scale0_groups(flow_ref_all, flow_ref)
    conf_ref_all, conf0 = scale0_groups(conf_ref_all, conf_ref)
    flows, confs = [flow0] + [None] * (t_scales - 1), [conf0] + [None] * (t_scales - 1)
    for s in range(1, t_scales):
        frames = real_B[s]
        if frames is None or frames.size(1) != tD:
            continue                                  # not enough history at this spacing yet
        later, earlier = frames[:, 1:], frames[:, :-1]
        flows[s], confs[s] = flowNet(later, earlier)
    return flow_ref_all, conf_ref_all, flows, confs


def _sparse_first_index(i, step):
    """Index inside the current chunk (whose first frame is frame i of the sequence) of the first frame that lies on the
    every-`step`-th grid anchored at frame 0 of the generated sequence (frame 0 = chunk 0, index 0; afterwards the grid
    continues from the previous chunk's frames, which end at global index i - 1)."""
    if i == 0:
        return 0
    return (step - 1) - ((i - 1) % step)


def get_skipped_frames_sparse(B_all, B, t_scales, tD, n_frames_load, i, is_flow=False):
    """--sparse_D bookkeeping (behaviour of the reference's :304-328, written from its specification).  Instead of one dense
    history that every temporal scale strides through, each scale s keeps only the frames it will ever look at (every
    tD**s-th frame), in groups of tD:
      * before appending, a history whose length is a whole number of groups is cut down to its last tD - 1 frames (the
        overlap the next group shares with it);
      * scale 0 appends the whole chunk, scale s >= 1 the chunk's frames that fall on its grid;
      * once at least tD frames are held, the oldest (length mod tD) frames are dropped and the rest is served as groups
        of tD; for flows / confidences the first entry of every group is dropped (tD frames <-> tD - 1 flows).
    B_all is the per-scale list of histories (updated in place and returned)."""
    ch, h, w = B.shape[2:]
    served = [None] * t_scales
    for s in range(t_scales):
        hist = B_all[s]
        if hist is not None and hist.size(1) > 0 and hist.size(1) % tD == 0:
            hist = hist[:, hist.size(1) - (tD - 1):]
        if s == 0:
            fresh = B
        else:
            step = tD ** s
            first = _sparse_first_index(i, step)
            fresh = B[:, first::step].contiguous() if first < n_frames_load else None
        if fresh is not None:
            hist = fresh if hist is None else torch.cat([hist.detach(), fresh], dim=1)
        held = 0 if hist is None else hist.size(1)
        if held >= tD:
            hist = hist[:, held % tD:]
            groups = hist.reshape(-1, tD, ch, h, w)
            served[s] = groups[:, 1:] if is_flow else groups
        B_all[s] = hist
    return B_all, served
