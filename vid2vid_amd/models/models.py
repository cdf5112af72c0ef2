"""Model factory of the MI355X backend (drop-in for the reference's models/models.py).

`create_model(opt)` returns `modelG` (test) or `[modelG, modelD, flowNet]` (train) exactly as
models/models.py:61-84 does; `create_optimizer` mirrors :86-102.  What differs is the parallel
runtime underneath: the reference wraps every model in nn.DataParallel and
re-broadcasts all parameters per call; here there is one process per GPU with persistent
replicas, wrapped in `RankModel`, which only exposes `.module` (the attribute train.py and
create_optimizer reach through) and all-reduces gradients over RCCL in `parallel.py`.
"""
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


def wrap_model(opt, modelG, modelD, flowNet):
    return RankModel(opt, modelG), RankModel(opt, modelD), RankModel(opt, flowNet)


def create_model(opt):
    print(opt.model)
    if opt.model != "vid2vid":
        raise ValueError("Model [%s] not recognized." % opt.model)
    from .vid2vid_model_G import Vid2VidModelG
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
        modelG, modelD, flowNet = wrap_model(opt, modelG, modelD, flowNet)
        return [modelG, modelD, flowNet]
    return modelG


def create_optimizer(opt, models):
    modelG, modelD, flowNet = models
    optimizer_G = modelG.module.optimizer_G
    optimizer_D = modelD.module.optimizer_D
    optimizer_D_T = [getattr(modelD.module, "optimizer_D_T" + str(s)) for s in range(opt.n_scales_temporal)]
    return modelG, modelD, flowNet, optimizer_G, optimizer_D, optimizer_D_T
