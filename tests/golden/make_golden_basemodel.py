#!/usr/bin/env python3
"""Vectors of the REFERENCE's BaseModel host helpers (models/base_model.py:146-181: get_edges, update_learning_rate,
update_training_batch) and Vid2VidModelG.compute_mask / compute_fake_B_prev (models/vid2vid_model_G.py:322-336), executed
unbound on plain namespaces.  Build container only.   python tests/golden/make_golden_basemodel.py"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG


def main():
    MG.install_shims()
    from models.base_model import BaseModel as RB
    from models.vid2vid_model_G import Vid2VidModelG as RG
    out = {"batch": [], "lr": []}
    for n_gpus in (1, 2, 6):
        for mfg in (1, 2, 4):
            for mfb in (1, 2, 4):
                s = types.SimpleNamespace(n_gpus=n_gpus, n_frames_per_gpu=1, n_frames_load=n_gpus, n_frames_bp=1,
                                          opt=types.SimpleNamespace(max_frames_per_gpu=mfg, max_frames_backpropagate=mfb))
                trace = []
                for ratio in range(0, 5):
                    RB.update_training_batch(s, ratio)
                    trace.append([s.n_frames_bp, s.n_frames_per_gpu, s.n_frames_load])
                out["batch"].append({"n_gpus": n_gpus, "max_frames_per_gpu": mfg, "max_frames_backpropagate": mfb, "trace": trace})
    for lr0, niter, decay in ((0.0002, 10, 10), (0.0001, 5, 20)):
        group = {"lr": lr0}
        s = types.SimpleNamespace(opt=types.SimpleNamespace(lr=lr0, niter=niter, niter_decay=decay), old_lr=lr0,
                                  optimizer_G=types.SimpleNamespace(param_groups=[group]))
        vals = []
        for epoch in range(niter + 1, niter + decay + 1, 3):
            RB.update_learning_rate(s, epoch, "G")
            vals.append([epoch, group["lr"], s.old_lr])
        out["lr"].append({"lr": lr0, "niter": niter, "niter_decay": decay, "vals": vals})
    gen = torch.Generator().manual_seed(900)
    inst = torch.randint(0, 3, (1, 2, 1, 3, 4), generator=gen).float().repeat_interleave(3, 3).repeat_interleave(2, 4)
    edges = RB.get_edges(types.SimpleNamespace(), inst)
    real_As = torch.rand(1, 4, 6, 5, 7, generator=gen)
    m1 = RG.compute_mask(types.SimpleNamespace(opt=types.SimpleNamespace(fg_labels=[2])), real_As, 1)
    m2 = RG.compute_mask(types.SimpleNamespace(opt=types.SimpleNamespace(fg_labels=[0, 3, 5])), real_As, 1, 3)
    rb_prev = torch.rand(1, 3, 3, 2, 2, generator=gen)
    fake = torch.rand(1, 3, 3, 2, 2, generator=gen)
    last = [torch.rand(1, 2, 3, 2, 2, generator=gen)]
    p1 = RG.compute_fake_B_prev(None, rb_prev, None, fake)
    p2 = RG.compute_fake_B_prev(None, rb_prev, last, fake)
    p3 = RG.compute_fake_B_prev(None, rb_prev, last, fake[:, :1])
    np.savez_compressed(os.path.join(HERE, "basemodel_helpers.npz"), inst=inst.numpy(), edges=edges.numpy(), real_As=real_As.numpy(),
                        m1=m1.numpy(), m2=m2.numpy(), rb_prev=rb_prev.numpy(), fake=fake.numpy(), last=last[0].numpy(),
                        p1=p1.numpy(), p2=p2.numpy(), p3=p3.numpy())
    json.dump(out, open(os.path.join(HERE, "basemodel_schedule.json"), "w"))
    print("wrote basemodel_helpers.npz, basemodel_schedule.json")


if __name__ == "__main__":
    main()
