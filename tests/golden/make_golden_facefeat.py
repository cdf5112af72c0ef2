#!/usr/bin/env python3
"""Vectors for the nearest-neighbour feature lookup of the REFERENCE's Vid2VidModelG.get_face_features
(models/vid2vid_model_G.py:290-320), executed with a stub encoder (the given instance-wise constant map) and a synthetic
features dictionary in place of the downloaded checkpoints/edge2face_single/features.npy.  Build container only.
    python tests/golden/make_golden_facefeat.py"""
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
    from models.vid2vid_model_G import Vid2VidModelG as RG
    from models.base_model import BaseModel as RB
    rs = np.random.RandomState(7)
    feat_num, H, W = 4, 16, 20
    n_img = {0: 9, 1: 12, 2: 9, 3: 10, 4: 9, 5: 11, 6: 9}       # part 6 sets num_images; the others hold at least as many rows
    features = {k: rs.uniform(-1, 1, size=(n, feat_num + 1)).astype(np.float32).astype(np.float64) for k, n in n_img.items()}  # float64 scalars are Python floats: the reference assigns them into tensor elements
    cases = []
    for case in range(3):
        parts = torch.from_numpy(rs.randint(0, 7, size=(1, 1, H // 4, W // 4))).float().repeat_interleave(4, 2).repeat_interleave(4, 3)
        parts[0, 0, 0, :7] = torch.arange(7).float()               # all seven parts present (absent rows are uninitialised there)
        target = rs.randint(0, 7)                                  # make one training image clearly the nearest
        feat_map = torch.zeros(1, feat_num, H, W)
        for lab in range(7):
            row = torch.from_numpy(features[lab][min(target, features[lab].shape[0] - 1), :feat_num]).float() + 0.01 * case
            feat_map[0].permute(1, 2, 0)[parts[0, 0] == lab] = row
        real = torch.zeros(1, 3, H, W)
        self = types.SimpleNamespace(netE=types.SimpleNamespace(forward=lambda img, inst, fm=feat_map: fm),
                                     opt=types.SimpleNamespace(feat_num=feat_num), Tensor=torch.FloatTensor,
                                     dists_min=lambda a, b, num=1: RB.dists_min(None, a, b, num))
        real_load = np.load
        np.load = lambda *a, **k: types.SimpleNamespace(item=lambda: features)
        try:
            out = RG.get_face_features(self, real, parts)
        finally:
            np.load = real_load
        cases.append((parts, feat_map, out.detach()))
    arrays = {"feat_num": np.array(feat_num)}
    for k, v in features.items():
        arrays["features.%d" % k] = v
    for i, (parts, fm, out) in enumerate(cases):
        arrays["case%d.inst" % i], arrays["case%d.feat_map" % i], arrays["case%d.out" % i] = parts.numpy(), fm.numpy(), out.numpy()
    np.savez_compressed(os.path.join(HERE, "face_feature_lookup.npz"), **arrays)
    print("wrote face_feature_lookup.npz", [float(c[2].abs().mean()) for c in cases])


if __name__ == "__main__":
    main()
