"""BaseModel helpers of the vid2vid model API on the MI355X backend.

Same helper surface as the reference's models/base_model.py (checkpoint naming and the
partial-match loading fallbacks :43-107, build_pyr :122-134, get_edges :146-152, concat
:109-120, lr / training-batch schedule :154-175, resample :183-196); tensor work goes
through libv2v_hip.so (engine.Engine) instead of torch ops.
"""
import os

import numpy as np
import torch

from .. import networks
from ..networks import get_engine


class _UnsteppedAdam:
    """What `self.optimizer_G` is after the reference's update_fixed_params (base_model.py:165): an Adam over all scales that
    nobody steps (train.py holds the optimizer it captured before).  Only its learning rate is ever touched again
    (update_learning_rate); step / zero_grad exist so that code written against the attribute does not break."""

    def __init__(self, lr, betas, n_params=0):
        self.param_groups = [{"lr": lr, "betas": tuple(betas), "params": []}]
        self.n_params = n_params
        self.grad_sync = None

    def zero_grad(self, set_to_none=False): pass
    def step(self, closure=None): return None
    def state_dict(self): return {}
    def load_state_dict(self, state): pass

    def rebuild(self, params, lr=None, betas=None):
        if lr is not None:
            self.param_groups[0]["lr"] = lr


class BaseModel(torch.nn.Module):
    def name(self):
        return "BaseModel"

    def initialize(self, opt):
        self.opt = opt
        self.gpu_ids = opt.gpu_ids
        self.isTrain = opt.isTrain
        if torch.cuda.is_available():
            self.device = torch.device("cuda", opt.gpu_ids[0] if len(opt.gpu_ids) else torch.cuda.current_device())
        elif networks._RECORD_ONLY["value"]:
            self.device = torch.device("cpu")      # plan-recording dry run (tests); cannot execute
        else:
            raise RuntimeError("vid2vid_amd needs an MI355X (no GPU visible and no CPU fallback exists)")
        self.dev_ids = [self.device.index] if self.device.type == "cuda" else []
        self.save_dir = os.path.join(opt.checkpoints_dir, opt.name)
        prec = getattr(opt, "precision", None)
        if prec is None:
            prec = "bf16" if getattr(opt, "fp16", False) else networks.get_precision()
        self.precision = prec
        networks.set_precision(prec)
        if self.isTrain:
            # the reference's norms run in training mode everywhere and update running_mean / running_var /
            # num_batches_tracked on every forward (networks.py:23-30); a checkpoint saved here must carry the same buffers
            self.engine.update_running_stats = True

    def Tensor(self, *shape):
        return torch.empty(*shape, dtype=torch.float32, device=self.device)

    @property
    def engine(self):
        # the model's OWN precision, not the process-wide default: an fp32 and a bf16 model may coexist
        return get_engine(self.device, networks._PREC_CODE[self.precision])

    def bind_precision(self):
        """Call at the end of initialize(): pins every network this model owns to the model's precision."""
        networks.bind_precision(self, self.precision)

    # ---------------- checkpoints: '<epoch>_net_<label>.pth' ----------------
    def _ckpt_path(self, label, epoch, save_dir=""):
        return os.path.join(save_dir or self.save_dir, "%s_net_%s.pth" % (epoch, label))

    def save_network(self, network, network_label, epoch_label, gpu_ids=None):
        os.makedirs(self.save_dir, exist_ok=True)
        from .. import parallel
        parallel.wait_pending()
        for m in network.modules():            # forwards counted on the host (engine._norm_params), written at save time
            nb = getattr(m, "_v2v_batches", 0)
            if nb and getattr(m, "num_batches_tracked", None) is not None:
                m.num_batches_tracked.add_(nb)
                m._v2v_batches = 0
        state = {k: v.detach().cpu().contiguous() for k, v in network.state_dict().items()}     # (flat-buffer weights are channels-last views)
        torch.save(state, self._ckpt_path(network_label, epoch_label))

    def load_network(self, network, network_label, epoch_label, save_dir=""):
        """Strict load, else 'checkpoint has extra layers' subset load, else per-key shape match
        (how 512 -> 1024 -> 2048 coarse-to-fine training re-uses weights); a missing G0 is an
        error unless opt.random_init_ok (benchmarks with random-init weights)."""
        path = self._ckpt_path(network_label, epoch_label, save_dir)
        if not os.path.isfile(path):
            print("%s not exists yet!" % path)
            if "G0" in network_label and not getattr(self.opt, "random_init_ok", False):
                raise RuntimeError("Generator must exist!")
            return
        saved = torch.load(path, map_location="cpu")
        try:
            network.load_state_dict(saved)
            return
        except RuntimeError:
            pass
        own = network.state_dict()
        try:
            network.load_state_dict({k: v for k, v in saved.items() if k in own})
            print("Pretrained network %s has excessive layers; Only loading layers that are used" % network_label)
            return
        except RuntimeError:
            pass
        print("Pretrained network %s has fewer layers; The following are not initialized:" % network_label)
        missing = set()
        for k, v in saved.items():
            if k in own and v.size() == own[k].size():
                own[k] = v
        for k, v in own.items():
            if k not in saved or v.size() != saved[k].size():
                missing.add(k.split(".")[0])
        print(sorted(missing))
        network.load_state_dict(own)

    # ---------------- tensor helpers ----------------
    def concat(self, tensors, dim=0):
        a, b = tensors
        if a is not None and b is not None:
            if isinstance(a, list):
                return [self.concat([x, y], dim=dim) for x, y in zip(a, b)]
            return torch.cat([a, b], dim=dim)
        return a if a is not None else b

    def build_pyr(self, tensor, nearest=False):
        """AvgPool2d(3,2,1,count_include_pad=False) pyramid over the two last dims."""
        if tensor is None:
            return [None] * self.n_scales
        pyr = [tensor]
        for _ in range(1, self.n_scales):
            t = pyr[-1].contiguous().float()
            if nearest:
                pyr.append(t[..., ::2, ::2].contiguous())
            elif self.engine.record_only:          # CPU dry run (plan recording only): shapes, no values
                pyr.append(t.new_zeros(*t.shape[:-2], (t.shape[-2] - 1) // 2 + 1, (t.shape[-1] - 1) // 2 + 1))
            else:
                pyr.append(self.engine.avgpool_planar(t))
        return pyr

    def get_edges(self, t):
        """Instance-boundary map; on the hot path this is fused into v2v_encode_labels."""
        edge = torch.zeros_like(t, dtype=torch.bool)
        dx = t[..., :, 1:] != t[..., :, :-1]
        dy = t[..., 1:, :] != t[..., :-1, :]
        edge[..., :, 1:] |= dx
        edge[..., :, :-1] |= dx
        edge[..., 1:, :] |= dy
        edge[..., :-1, :] |= dy
        return edge.float()

    def resample(self, image, flow):
        return self.engine.resample_flow(image.contiguous().float(), flow.contiguous().float())

    # ---------------- schedules ----------------
    def update_learning_rate(self, epoch, model):
        lr = self.opt.lr * (1 - (epoch - self.opt.niter) / self.opt.niter_decay)
        for group in getattr(self, "optimizer_" + model).param_groups:
            group["lr"] = lr
        print("update learning rate: %f -> %f" % (self.old_lr, lr))
        self.old_lr = lr

    def update_fixed_params(self):
        """reference models/base_model.py:161-168, called by update_models / init_params once epoch > niter_fix_global.

        What the reference's call DOES, observably: it builds a NEW torch.optim.Adam over all scales and stores it in
        `self.optimizer_G` -- but train.py stepped, and keeps stepping, the optimizer object it captured at start-up
        (train.py:29: create_optimizer runs before init_params).  So after this call (a) the finest scale goes on training with
        its old moments, (b) the coarse scales receive gradients (`finetune_all` stops the detach of :181-186) that no
        optimizer ever applies -- their weights never move -- and (c) update_learning_rate (:154-159) from now on decays the
        learning rate of the new, never-stepped object while the captured one keeps the rate it had.

        Default here = exactly that behaviour, so that a run switched over from the reference produces the reference's
        parameters: the captured FusedAdam is left alone (`self._optimizer_G_live` keeps it reachable from the model -- e.g. for
        `FusedAdam.state_dict()` -- nothing in the package reads it), `self.optimizer_G` becomes an unstepped stand-in that takes (c), and the coarse-scale gradients of (b), which
        nothing can observe, are not computed (`_train_coarse` stays False: the finest scale's gradient is identical with or
        without the detach).  `opt.fix_update_fixed_params` (or V2V_FIX_UPDATE_FIXED_PARAMS=1) selects what the reference's
        authors evidently meant instead: the captured optimizer is rebuilt IN PLACE over all scales (fresh moments, lr /
        betas of the reference's new Adam), so train.py's handle trains every scale.  INTEGRATION.md section A documents both."""
        params = []
        for s in range(self.n_scales):
            params += list(getattr(self, "netG" + str(s)).parameters())
        fix = bool(getattr(self.opt, "fix_update_fixed_params", False)) or os.environ.get("V2V_FIX_UPDATE_FIXED_PARAMS", "0") == "1"
        if fix:
            self.optimizer_G.rebuild(params, lr=self.old_lr, betas=(self.opt.beta1, 0.999))
            self._train_coarse = True
        elif not isinstance(self.optimizer_G, _UnsteppedAdam):
            self._optimizer_G_live = self.optimizer_G
            self.optimizer_G = _UnsteppedAdam(lr=self.old_lr, betas=(self.opt.beta1, 0.999), n_params=sum(p.numel() for p in params))
            print("vid2vid_amd: WARNING -- reference-compatible update_fixed_params (train.py keeps stepping the optimizer it captured "
                  "at start-up): the coarse scales stay frozen and the generator's learning rate no longer decays from here on.  "
                  "Set --fix_update_fixed_params (or V2V_FIX_UPDATE_FIXED_PARAMS=1) to train every scale, which is what the "
                  "message below promises.")
        self.finetune_all = True
        print("------------ Now finetuning all scales -----------")

    def update_training_batch(self, ratio):
        nfb, nfl = self.n_frames_bp, self.n_frames_load
        if nfb < nfl:
            nfb = min(self.opt.max_frames_backpropagate, 2 ** ratio)
            self.n_frames_bp = nfl // int(np.ceil(float(nfl) / nfb))
            print("-------- Updating number of backpropagated frames to %d ----------" % self.n_frames_bp)
        if self.n_frames_per_gpu < self.opt.max_frames_per_gpu:
            self.n_frames_per_gpu = min(self.n_frames_per_gpu * 2, self.opt.max_frames_per_gpu)
            self.n_frames_load = self.n_gpus * self.n_frames_per_gpu
            print("-------- Updating number of frames per gpu to %d ----------" % self.n_frames_per_gpu)
