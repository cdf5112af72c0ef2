#!/usr/bin/env python3
"""Golden vectors for the on-GPU visualisation kernels: the REFERENCE's own util.tensor2im / util.tensor2label
(util/util.py:48-87, Colorize :197-212, labelcolormap :156-181) run here on CPU on seeded inputs.
    python tests/golden/make_golden_visual.py      (needs /root/reference; writes tests/golden/visual_util.npz)"""
import os
import sys
import types

import numpy as np
import torch

sys.path.insert(0, "/root/reference")
sys.modules.setdefault("cv2", types.ModuleType("cv2"))
import util.util as U      # noqa: E402

torch.manual_seed(5)
out = {}
img = torch.randn(1, 3, 24, 40) * 0.8                      # values beyond [-1, 1]: exercises the clip
img[0, 0, 0, :4] = torch.tensor([-1.0, 1.0, 0.0, 0.999])
out["im.x"] = img.numpy()
out["im.norm"] = U.tensor2im(img)
out["im.raw"] = U.tensor2im(img.abs(), normalize=False)
w = torch.rand(1, 1, 24, 40)                                # single plane (the `weight` map): (H, W) output
out["w.x"] = w.numpy()
out["w.raw"] = U.tensor2im(w, normalize=False)
seq = torch.randn(1, 2, 3, 8, 12)                           # 5-D: [0, -1]
out["seq.x"] = seq.numpy()
out["seq.norm"] = U.tensor2im(seq)
for n in (35, 20, 12):                                      # the two Cityscapes tables and the generated map
    lab = torch.randint(0, n, (24, 40))
    onehot = torch.zeros(n + 1, 24, 40)                     # + the instance-edge plane of real_A
    onehot.scatter_(0, lab.unsqueeze(0), 1.0)
    onehot[n] = (torch.rand(24, 40) < 0.1).float()
    onehot[lab[3, 5], 3, 5] = 0.0                           # an all-zero column apart from nothing: argmax -> first index
    out["lab%d.x" % n] = onehot.numpy()
    out["lab%d.rgb" % n] = U.tensor2label(onehot, n)
    out["lab%d.cmap" % n] = U.labelcolormap(n)
    ids = lab.float().unsqueeze(0)                          # single plane of ids
    out["lab%d.ids" % n] = ids.numpy()
    out["lab%d.ids_rgb" % n] = U.tensor2label(ids, n)
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "visual_util.npz"), **out)
print({k: v.shape for k, v in out.items()})
