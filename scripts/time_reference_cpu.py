#!/usr/bin/env python3
"""The REFERENCE's own classes timed on this container's host cores (SURVEY.md Appendix B recipe; nothing under /root/reference is
modified or copied): Vid2VidModelG.inference of NVIDIA/vid2vid on CPU, label2city, --fg --use_instance, random-init weights, at
BASELINE configs[0] (256x128, 2-frame clip) and at the headline geometry (512x256).  This is the `kind: "reference"` counterpart of
bench.py's `cpu_baseline` (`kind: "port"`, the oracle restatement timed on the GPU box's host, where /root/reference does not
exist); it can only be taken here.      python scripts/time_reference_cpu.py > profiles/r06_reference_cpu_timing.txt"""
import os
import sys
import time
import types

import torch

sys.path.insert(0, "/root/reference")
for m in ["torchvision", "torchvision.models", "cv2", "dominate", "dominate.tags", "scipy.misc"]:
    sys.modules.setdefault(m, types.ModuleType(m))
sys.modules["torchvision"].models = sys.modules["torchvision.models"]
torch.Tensor.cuda = lambda self, *a, **k: self
torch.cuda.FloatTensor = torch.FloatTensor
torch.cuda.ByteTensor = torch.ByteTensor
os.makedirs("/tmp/ckpt_ref/probe", exist_ok=True)


def run(W, H, frames):
    sys.argv = ["test.py", "--name", "probe", "--label_nc", "35", "--loadSize", str(W), "--use_instance", "--fg", "--use_real_img",
                "--gpu_ids", "-1", "--checkpoints_dir", "/tmp/ckpt_ref"]
    from options.test_options import TestOptions
    opt = TestOptions().parse(save=False)
    from models import networks
    torch.manual_seed(0)
    netG0 = networks.define_G(35 * 3 + 3, 3, 6, opt.ngf, "composite", opt.n_downsample_G, opt.norm, 0, [], opt)
    torch.save(netG0.state_dict(), "/tmp/ckpt_ref/probe/latest_net_G0.pth")
    from models.models import create_model
    model = create_model(opt)
    g = torch.Generator().manual_seed(1)
    tG = 3
    lab = torch.randint(0, 35, (1, frames + tG - 1, 1, H, W), generator=g).float()
    inst = torch.randint(0, 20, (1, frames + tG - 1, 1, H, W), generator=g).float()
    B = torch.tanh(torch.randn(1, tG - 1, 3, H, W, generator=g))
    times = []
    with torch.no_grad():
        for t in range(frames):
            t0 = time.perf_counter()
            model.inference(lab[:, t:t + tG], B if t == 0 else None, inst[:, t:t + tG])
            times.append(time.perf_counter() - t0)
    return times


if __name__ == "__main__":
    name = ""
    for line in open("/proc/cpuinfo"):
        if line.startswith("model name"):
            name = line.split(":", 1)[1].strip(); break
    print("# NVIDIA/vid2vid reference classes (models/vid2vid_model_G.py Vid2VidModelG.inference) on the build container's host: %d torch threads, %s"
          % (torch.get_num_threads(), name))
    for (W, H, n) in ((256, 128, 3), (512, 256, 2)):
        _stdout = sys.stdout
        sys.stdout = sys.stderr
        try:
            ts = run(W, H, n)
        finally:
            sys.stdout = _stdout
        steady = ts[1:]
        print("label2city %dx%d, n_scales_spatial=1, ngf=128, --fg --use_instance, fp32: frame times %s s -> %.4f frames/s (frames after the first; first frame %.2f s)"
              % (W, H, ["%.2f" % t for t in ts], len(steady) / sum(steady), ts[0]))
