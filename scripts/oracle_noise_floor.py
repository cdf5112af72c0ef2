#!/usr/bin/env python3
"""Noise floor of the training-chunk gradient metric.  CPU only; TEST INFRASTRUCTURE (imports oracle/).

bench.py / tests gate the fp32 HIP chunk against the fp32 CPU oracle by the relative L2 distance of the whole flattened
gradient.  Both sides are fp32 evaluations of the same function in different summation orders.  How far apart may two CORRECT
fp32 evaluations be?  This script runs the oracle chunk twice on the same weights and inputs -- in fp32 and in fp64 -- and
reports the fp32 oracle's own distance from the fp64 result with the same measure (norm error, L2 distance, per-tensor
attribution).  An fp32 implementation cannot be expected to sit closer to the fp32 oracle than ~sqrt(2) x this distance.

    python scripts/oracle_noise_floor.py --width 256 --height 128 --frames 3 [--ngf 128] > profiles/r05_oracle_noise_floor_256x128.json
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch
import torch.nn.functional as F


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=256)
    ap.add_argument("--height", type=int, default=128)
    ap.add_argument("--frames", type=int, default=3)
    ap.add_argument("--ngf", type=int, default=128)
    ap.add_argument("--num-D", type=int, default=2)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--vgg", action="store_true", help="with the VGG19 perceptual loss (seeded stand-in weights of tests/util.py), as bench.py's train leg runs it")
    args = ap.parse_args()
    if args.threads:
        torch.set_num_threads(args.threads)
    from oracle import train_parity as TP
    from vid2vid_amd import networks as N, synthetic
    N.set_record_only(True)
    from vid2vid_amd.options import make_opt
    H, W, nfl = args.height, args.width, args.frames
    torch.manual_seed(0)
    opt = make_opt(isTrain=True, label_nc=35, loadSize=W, use_instance=True, fg=True, n_scales_spatial=1, num_D=args.num_D, no_vgg=True,
                   n_frames_total=max(nfl, 2) * 2, max_frames_per_gpu=nfl, n_scales_temporal=1, niter_fix_global=0, precision="fp32",
                   gpu_ids=[0], random_init_ok=True, ngf=args.ngf)
    from vid2vid_amd.models.vid2vid_model_G import Vid2VidModelG
    from vid2vid_amd.models.vid2vid_model_D import Vid2VidModelD
    G = Vid2VidModelG(); G.initialize(opt)               # record-only backend: the networks are plain nn.Modules on the CPU
    D = Vid2VidModelD(); D.initialize(opt)
    netG, netD, netDT = G.netG0, D.netD, D.netD_T0
    with torch.no_grad():
        netG.model_final_flow[1].weight.mul_(0.1)
    tG = opt.n_frames_G
    nT = nfl + tG - 1
    lab, inst, frames = synthetic.label2city_sequence(nT, H, W, seed=1234, device="cpu")
    gen = torch.Generator().manual_seed(99)
    flow_ref = torch.randn(1, nfl, 2, H // 8, W // 8, generator=gen) * 2.0
    flow_ref = F.interpolate(flow_ref.view(-1, 2, H // 8, W // 8), size=(H, W), mode="bilinear", align_corners=False).view(1, nfl, 2, H, W)
    conf_ref = (torch.rand(1, nfl, 1, H // 4, W // 4, generator=gen) > 0.3).float().repeat_interleave(4, 3).repeat_interleave(4, 4)
    sd = lambda m: {k: v.detach().clone() for k, v in m.state_dict().items()}
    names = {"G": [[n for n, p in netG.named_parameters() if p.requires_grad]],
             "D": [n for n, p in netD.named_parameters() if p.requires_grad],
             "DT": [n for n, p in netDT.named_parameters() if p.requires_grad]}
    has_T = nfl >= opt.n_frames_D
    sd_vgg = None
    if args.vgg:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from util import seeded_vgg19_features, vgg19_slice_state_dict
        sd_vgg = vgg19_slice_state_dict(seeded_vgg19_features(77))

    def run(dtype):
        torch.set_default_dtype(dtype)
        try:
            cast = lambda d: {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in d.items()}
            t0 = time.perf_counter()
            out = TP.oracle_chunk([cast(sd(netG))], cast(sd(netD)), cast(sd(netDT)) if has_T else None,
                                  lab.view(1, nT, 1, H, W).to(dtype), inst.view(1, nT, 1, H, W).to(dtype), frames.to(dtype),
                                  flow_ref.to(dtype), conf_ref.to(dtype), n_down=opt.n_downsample_G, n_blocks=opt.n_blocks,
                                  n_blocks_local=opt.n_blocks_local, n_frames_load=nfl, num_D=args.num_D,
                                  sd_vgg=None if sd_vgg is None else cast(sd_vgg), param_names=names,
                                  dtype=dtype)
            out["seconds"] = time.perf_counter() - t0
            return out
        finally:
            torch.set_default_dtype(torch.float32)

    r64 = run(torch.float64)
    r32 = run(torch.float32)
    r32["outs"] = {k: v.double() for k, v in r32["outs"].items()}
    cmp = TP.compare(r32, r64)
    rec = {"what": "fp32 CPU oracle vs the SAME oracle in fp64 (same weights, same inputs): the distance two correct fp32 evaluations of this "
                   "training chunk may have from the exact result, in the measures bench.py / tests gate the HIP path with",
           "chunk": "label2city %dx%d, ngf=%d, %d frames, num_D=%d, --no_vgg, temporal scale 0 %s" % (W, H, args.ngf, nfl, args.num_D, "active" if has_T else "inactive"),
           "threads": torch.get_num_threads(), "seconds": {"fp64": round(r64["seconds"], 1), "fp32": round(r32["seconds"], 1)},
           "forward": cmp["forward"], "losses": cmp["losses"], "grads": cmp["grads"], "grad_error_by_tensor": cmp.get("grad_error_by_tensor")}
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
