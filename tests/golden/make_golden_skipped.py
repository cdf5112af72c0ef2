#!/usr/bin/env python3
"""Golden vectors for the temporal-discriminator frame bookkeeping: executes the REFERENCE's own get_skipped_frames /
get_skipped_flows / get_skipped_frames_sparse (models/vid2vid_model_D.py:274-328) on index-valued tiny tensors over several
chunks of a sequence.  Build container only.     python tests/golden/make_golden_skipped.py
Frames are (1, n, 1, 1, 1) tensors holding their own frame index, so the fixture records WHICH frames each temporal scale
receives; the stub flowNet is f(a, b) = (a - b/2, a + b/4)."""
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG

CASES = [(1, 3, 1), (2, 3, 1), (2, 3, 2), (3, 3, 3), (2, 2, 1), (3, 2, 2), (2, 3, 4)]     # t_scales, tD, n_frames_load
N_CHUNKS = 14


def flownet_stub(a, b):
    return a - b / 2, a + b / 4


def tolist(t):
    return None if t is None else [list(map(float, row)) for row in t.reshape(t.shape[0], -1).tolist()]


def main():
    MG.install_shims()
    from models import vid2vid_model_D as R
    out = {"cases": []}
    for (t_scales, tD, nfl) in CASES:
        rec = {"t_scales": t_scales, "tD": tD, "n_frames_load": nfl, "dense": [], "sparse": []}
        # dense
        real_all = flow_all = conf_all = None
        for c in range(N_CHUNKS):
            fr = torch.arange(c * nfl, (c + 1) * nfl, dtype=torch.float32).view(1, nfl, 1, 1, 1)
            real_all, real_sk = R.get_skipped_frames(real_all, fr, t_scales, tD)
            flow_all, conf_all, flow_sk, conf_sk = R.get_skipped_flows(flownet_stub, flow_all, conf_all, real_sk, fr * 10, fr * 100,
                                                                      t_scales, tD)
            rec["dense"].append({"all": tolist(real_all), "sk": [tolist(t) for t in real_sk],
                                 "flow_all": tolist(flow_all), "flow_sk": [tolist(t) for t in flow_sk],
                                 "conf_sk": [tolist(t) for t in conf_sk]})
        # sparse
        b_all, f_all = [None] * t_scales, [None] * t_scales
        for c in range(N_CHUNKS):
            i = c * nfl
            fr = torch.arange(i, i + nfl, dtype=torch.float32).view(1, nfl, 1, 1, 1)
            b_all, b_sk = R.get_skipped_frames_sparse(b_all, fr, t_scales, tD, nfl, i)
            f_all, f_sk = R.get_skipped_frames_sparse(f_all, fr * 10, t_scales, tD, nfl, i, is_flow=True)
            rec["sparse"].append({"all": [tolist(t) for t in b_all], "sk": [tolist(t) for t in b_sk],
                                  "flow_sk": [tolist(t) for t in f_sk]})
        out["cases"].append(rec)
    path = os.path.join(HERE, "skipped_frames.json")
    json.dump(out, open(path, "w"))
    print("wrote", path, os.path.getsize(path) // 1024, "KB")


if __name__ == "__main__":
    main()
