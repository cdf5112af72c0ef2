#!/usr/bin/env python3
"""Golden fixture for the VGG19 perceptual loss: executes the REFERENCE's own `VGGLoss` / `Vgg19` classes
(models/networks.py:776-791, 840-870) on CPU.  Build container only (needs /root/reference).

    python tests/golden/make_golden_vgg.py        # writes tests/golden/vgg_loss_32x64.npz

torchvision is an external dependency of the reference that is not installed here, and its pretrained vgg19 weights
are a download: `torchvision.models.vgg19(pretrained=True)` is stubbed by the published configuration-'E' feature
stack (Conv3x3+ReLU / MaxPool2d(2,2)) carrying tests/util.seeded_vgg19_features -- a host-independent seeded stand-in
that the tests regenerate (12.9 M values are not stored).  What the fixture pins is therefore the reference's slicing
(indices 2/7/12/21/30), the slice weights 1/32..1, L1 + detach, and the gradient through it -- on real VGG19 widths.
"""
import os
import sys

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import make_golden as MG                      # shims
from util import seeded_vgg19_features


def vgg19_stub(pretrained=True):
    cfg = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512, "M"]
    mods, cin = [], 3
    for v in cfg:
        if v == "M":
            mods.append(nn.MaxPool2d(kernel_size=2, stride=2))
        else:
            mods += [nn.Conv2d(cin, v, 3, padding=1), nn.ReLU(inplace=True)]
            cin = v
    net = nn.Module()
    net.features = nn.Sequential(*mods)
    sd = seeded_vgg19_features(upto=37)
    net.load_state_dict(sd)
    return net


def main():
    MG.install_shims()
    sys.modules["torchvision.models"].vgg19 = vgg19_stub
    from models import networks as R
    crit = R.VGGLoss(0)
    gen = torch.Generator().manual_seed(404)
    n, H, W = 2, 32, 64
    up = lambda t: torch.tanh(torch.nn.functional.interpolate(t, size=(H, W), mode="bilinear", align_corners=False))
    y = up(torch.randn(n, 3, H // 4, W // 4, generator=gen))
    x = (y + 0.3 * up(torch.randn(n, 3, H // 2, W // 2, generator=gen))).clamp(-1, 1).requires_grad_(True)
    loss = crit(x, y)
    loss.backward()
    feats = crit.vgg(x.detach())
    arrays = {"in.x": x.detach().numpy(), "in.y": y.numpy(), "out.loss": np.array(float(loss)),
              "out.grad_x": x.grad.numpy(), "seed": np.array(77)}
    for i, f in enumerate(feats):
        arrays["out.feat%d" % (i + 1)] = f.detach().numpy()
    MG.save("vgg_loss_32x64", **arrays)
    print("loss", float(loss), "grad rms", float(x.grad.pow(2).mean().sqrt()),
          "feat rms", [float(f.pow(2).mean().sqrt()) for f in feats])

    # the >1024-wide branch (VGGLoss.downsample, :782-786): loss only
    xw = torch.tanh(torch.randn(1, 3, 64, 1280, generator=gen))
    yw = torch.tanh(torch.randn(1, 3, 64, 1280, generator=gen))
    MG.save("vgg_loss_wide_64x1280", **{"in.x": xw.numpy().astype(np.float16), "in.y": yw.numpy().astype(np.float16),
                                        "out.loss": np.array(float(crit(xw.half().float(), yw.half().float())))})


if __name__ == "__main__":
    main()
