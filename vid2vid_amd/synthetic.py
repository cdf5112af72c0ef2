"""Seeded synthetic label / instance / image sequences; no dataset,
no network.  Shared by bench.py, __graft_entry__.smoke() and the full-size parity tests."""
import torch
import torch.nn.functional as F


def label2city_sequence(T, H, W, seed=1234, n_labels=35, n_inst=20, fg_label=26, cell=16, device="cpu"):
    """Returns (labels (T,H,W), inst (T,H,W), frames (1,T,3,H,W)); labels/inst are floats holding
    integers (Appendix C.10), frames are in [-1, 1].  Temporally coherent: 2 px/frame roll in x."""
    gen = torch.Generator().manual_seed(seed)
    h, w = max(H // cell, 1), max(W // cell, 1)
    lo = torch.randint(0, n_labels, (h, w), generator=gen)
    fg_cells = torch.rand(h, w, generator=gen) < 0.05
    fg_cells[0, 0] = True
    lo[fg_cells] = fg_label
    lab0 = lo.repeat_interleave(cell, 0).repeat_interleave(cell, 1)[:H, :W]
    gen_i = torch.Generator().manual_seed(seed + 1)
    io = torch.randint(0, n_inst, (h, w), generator=gen_i)
    inst0 = io.repeat_interleave(cell, 0).repeat_interleave(cell, 1)[:H, :W]
    gen_b = torch.Generator().manual_seed(seed + 2)
    img0 = torch.tanh(F.interpolate(torch.randn(1, 3, max(H // 8, 1), max(W // 8, 1), generator=gen_b),
                                    size=(H, W), mode="bilinear", align_corners=False))[0]
    labs, insts, imgs = [], [], []
    for t in range(T):
        labs.append(torch.roll(lab0, 2 * t, dims=1))
        insts.append(torch.roll(inst0, 2 * t, dims=1))
        imgs.append(torch.roll(img0, 2 * t, dims=2))
    lab = torch.stack(labs).float().to(device)
    inst = torch.stack(insts).float().to(device)
    frames = torch.stack(imgs).unsqueeze(0).to(device)
    return lab, inst, frames


def edge2face_sequence(T, H, W, seed=1234, input_nc=15, device="cpu"):
    """Returns (A (1,T,input_nc,H,W), frames (1,T,3,H,W)) for the edge2face geometry: channel 0 sparse
    binary edges (p = 0.02), channels 1.. smooth maps in [0,1]; 2 px/frame roll in x."""
    gen = torch.Generator().manual_seed(seed)
    edges = (torch.rand(1, H, W, generator=gen) < 0.02).float()
    smooth = torch.sigmoid(F.interpolate(torch.randn(1, input_nc - 1, max(H // 16, 1), max(W // 16, 1), generator=gen),
                                         size=(H, W), mode="bilinear", align_corners=False))[0]
    a0 = torch.cat([edges, smooth], 0)
    img0 = torch.tanh(F.interpolate(torch.randn(1, 3, max(H // 8, 1), max(W // 8, 1), generator=gen),
                                    size=(H, W), mode="bilinear", align_corners=False))[0]
    A = torch.stack([torch.roll(a0, 2 * t, dims=2) for t in range(T)]).unsqueeze(0).contiguous().to(device)
    frames = torch.stack([torch.roll(img0, 2 * t, dims=2) for t in range(T)]).unsqueeze(0).contiguous().to(device)
    return A, frames
