"""Model factory of the MI355X backend (drop-in for the reference's models/models.py).

`create_model(opt)` returns `modelG` (test) or `[modelG, modelD, flowNet]` (train) exactly as
models/models.py:61-84 does; `create_optimizer` mirrors :86-102.  What differs is the parallel
runtime underneath: the reference wraps every model in nn.DataParallel and
re-broadcasts all parameters per call; here there is one process per GPU with persistent
replicas, wrapped in `RankModel`, which only exposes `.module` (the attribute train.py and
create_optimizer reach through) and all-reduces gradients over RCCL in `parallel.py`.
"""
import os
import torch
import torch.nn as nn


class RankModel(nn.Module):
    """Per-rank stand-in for the reference's myModel / nn.DataParallel wrappers
    (models/models.py:10-59): same call signature, `.module` attribute, no replication."""

    def __init__(self, opt, model):
        super().__init__()
        self.opt = opt
        self.module = model

    def forward(self, *inputs, **kwargs):
        from .. import parallel
        parallel.wait_pending()          # optimizer steps overlapped with the previous backward passes (parallel.GradSync)
        return self.module(*inputs, **kwargs)


def wrap_model(opt, modelG, modelD, flowNet, layout=None):
    """reference models/models.py:10-23.  `--n_gpus_gen` smaller than the number of GPUs of a sequence group selects the
    generator / discriminator rank roles (vid2vid_amd/roles.py); otherwise every rank trains G and D on its own sequence."""
    if layout is not None:
        from ..roles import wrap_roles
        return wrap_roles(opt, modelG, modelD, flowNet, layout)
    return RankModel(opt, modelG), RankModel(opt, modelD), RankModel(opt, flowNet)


def _role_layout(opt):
    """One process per GPU: `--gpu_ids` (the reference's list of the GPUs that share ONE sequence) becomes the size of a
    sequence group and this process keeps its own device only."""
    import os
    from .. import roles
    layout = roles.layout_from_opt(opt) if opt.isTrain else None
    if layout is not None:
        opt.role_group_size = layout.group_size
        # one process per GPU: this process's device is the launcher's LOCAL_RANK (what parallel.init_distributed / bench.py
        # selected with torch.cuda.set_device), NOT an index into --gpu_ids -- two sequence groups on one node (8 processes,
        # --gpu_ids 0,1,2,3) would otherwise put ranks 4-7 on the GPUs of ranks 0-3
        # LOCAL_RANK unset (srun / mpirun launchers): the global rank modulo the GPUs of this node -- the global rank itself would
        # be out of range on every node after the first.  The user's --gpu_ids values are NOT a device map here (they
        # only give the size of a sequence group): restrict / reorder devices with HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES.
        ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
        local = int(os.environ["LOCAL_RANK"]) if "LOCAL_RANK" in os.environ else (layout.rank % ndev if ndev else layout.rank)
        if torch.cuda.is_available():
            if local >= ndev:
                raise RuntimeError("role split: LOCAL_RANK %d but only %d GPUs are visible (one process per GPU)" % (local, ndev))
            torch.cuda.set_device(local)
        opt.gpu_ids = [local]
        print("vid2vid_amd: rank %d = %s-rank %d of sequence group %d (%d generator + %d discriminator ranks per sequence)"
              % (layout.rank, layout.role, layout.g if layout.role == "G" else layout.d, layout.seq, layout.n_gen, layout.n_disc))
    return layout


def create_model(opt):
    print(opt.model)
    if opt.model != "vid2vid":
        raise ValueError("Model [%s] not recognized." % opt.model)
    from .vid2vid_model_G import Vid2VidModelG
    layout = _role_layout(opt)
    modelG = Vid2VidModelG()
    if opt.isTrain:
        from .vid2vid_model_D import Vid2VidModelD
        from .flownet import FlowNet
        modelD = Vid2VidModelD()
        flowNet = FlowNet()
    modelG.initialize(opt)
    if opt.isTrain:
        modelD.initialize(opt)
        flowNet.initialize(opt)
        modelG, modelD, flowNet = wrap_model(opt, modelG, modelD, flowNet, layout)
        return [modelG, modelD, flowNet]
    return modelG


def create_optimizer(opt, models):
    modelG, modelD, flowNet = models
    optimizer_G = modelG.module.optimizer_G
    optimizer_D = modelD.module.optimizer_D
    optimizer_D_T = [getattr(modelD.module, "optimizer_D_T" + str(s)) for s in range(opt.n_scales_temporal)]
    # the caller is train.py's loop: zero_grad() -> backward() -> step() per optimizer.  The discriminators' weight gradients of
    # loss_G.backward() are wiped by optimizer_D.zero_grad() before anything reads them; V2V_DISCARD_STALE_GRADS=1 does not compute
    # them (optim.FusedAdam.discard_stale_grads; bit-identical weights, tests/test_gpu_train_ops.py).  Measured: 37.7 / 38.4 vs
    # 38.4 / 37.9 frames trained/s (profiles/r06_v40_trainab.txt) -- nothing, so torch's exact semantics stay the default.
    if os.environ.get("V2V_DISCARD_STALE_GRADS", "0") == "1":
        for o in [optimizer_G, optimizer_D] + optimizer_D_T:
            if hasattr(o, "discard_stale_grads"):
                o.discard_stale_grads = True
    return modelG, modelD, flowNet, optimizer_G, optimizer_D, optimizer_D_T
