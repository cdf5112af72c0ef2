#!/usr/bin/env python3
"""Golden fixtures for the first-frame generators with instance-wise feature encoding: executes the
REFERENCE's own Global_with_z / Local_with_z / Encoder classes (models/networks.py:421-632) on CPU.  Build container only.

    python tests/golden/make_golden_face.py       # writes tests/golden/face_first_frame_nets_32x32.npz
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG


def main():
    MG.install_shims()
    from models import networks as R
    arrays = {}
    gen = torch.Generator().manual_seed(500)
    H = W = 32
    nz = 4          # face recipe: feat_num = 16; the HIP norm kernels need channel counts that are multiples of 4
    x = torch.rand(1, 5, H, W, generator=gen)
    z = torch.tanh(torch.randn(1, nz, H, W, generator=gen))
    # instance map: blocky ids incl. a large id (Cityscapes-style 26001) and a single-pixel instance
    inst = torch.randint(0, 4, (1, 1, H // 8, W // 8), generator=gen).float().repeat_interleave(8, 2).repeat_interleave(8, 3)
    inst[0, 0, :8, :8] = 26001.0
    inst[0, 0, 31, 31] = 7.0

    torch.manual_seed(501)
    opt = MG.opt_ns(n_blocks=2, feat_num=nz)
    g = R.define_G(5, 3, 0, 8, "global_with_features", 2, "instance", 0, [], opt)
    with torch.no_grad():
        arrays["out.global_with_z"] = g(x, z).numpy()
    arrays.update(MG.sd_to_np(g.state_dict(), "sdG."))

    torch.manual_seed(502)
    opt = MG.opt_ns(n_blocks=2, n_local_enhancers=1, n_blocks_local=1, feat_num=nz)
    l = R.define_G(5, 3, 0, 4, "local_with_features", 2, "instance", 0, [], opt)
    with torch.no_grad():
        arrays["out.local_with_z"] = l(x, z).numpy()
    arrays.update(MG.sd_to_np(l.state_dict(), "sdL."))

    torch.manual_seed(503)
    e = R.define_G(3, nz, 0, 4, "encoder", 2, "instance", 0, [])
    img = torch.tanh(torch.randn(1, 3, H, W, generator=gen))
    with torch.no_grad():
        arrays["out.encoder"] = e(img, inst).numpy()
    arrays.update(MG.sd_to_np(e.state_dict(), "sdE."))
    arrays.update({"in.x": x.numpy(), "in.z": z.numpy(), "in.inst": inst.numpy(), "in.img": img.numpy()})
    MG.save("face_first_frame_nets_32x32", **arrays)


if __name__ == "__main__":
    main()
