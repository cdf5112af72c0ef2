#!/usr/bin/env python3
"""Runs the REFERENCE's own train.py / test.py (from /root/reference, unmodified) against the MI355X backend through the
three-line `models/models.py` shim of INTEGRATION.md section A, on a host without a GPU.

What is real: the reference's driver scripts, option parsers (options/*.py), Visualizer / util helpers, and the tensor
contract of its BaseDataset (init_data_params / init_data / prepare_data, data/base_dataset.py:56-80).  What is replaced:
`models.models` (the shim), `data.data_loader` (a synthetic dataset that subclasses the reference's BaseDataset: image
folders / PIL loading are out of scope), and import stubs for third-party packages absent here.
The backend runs in dry-run mode (include/v2v_hip.h, v2v_set_dry_run): every launch is argument-checked by the library,
autograd graphs are built and walked backward, optimizers step -- nothing executes, so tensor VALUES are meaningless;
the point is that every attribute, call, shape and return structure the reference drivers rely on exists.

    python tests/dropin/run_reference.py train|test <workdir>      -> prints one JSON report line prefixed DROPIN_REPORT
"""
import json
import os
import runpy
import sys
import types

mode, work = sys.argv[1], sys.argv[2]
REF = "/root/reference"
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:] = [p for p in sys.path if os.path.abspath(p or ".") != os.path.join(REPO, "tests")]     # tests/util.py would shadow the reference's util package
sys.path.insert(0, REF)
sys.path.insert(1, REPO)

import torch

for m in ["torchvision", "torchvision.models", "torchvision.transforms", "cv2", "dominate", "dominate.tags", "scipy.misc"]:
    sys.modules.setdefault(m, types.ModuleType(m))
sys.modules["torchvision"].models = sys.modules["torchvision.models"]
sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]

from vid2vid_amd import networks as N
N.set_record_only(True)

# ---- the shim that replaces the reference's models/models.py ----
shim = types.ModuleType("models.models")
exec("from vid2vid_amd.models.models import create_model, create_optimizer, wrap_model\n"
     "from vid2vid_amd.models.schedule import init_params, save_models, update_models\n", shim.__dict__)
pkg = types.ModuleType("models")
pkg.__path__ = []
pkg.models = shim
sys.modules["models"], sys.modules["models.models"] = pkg, shim

report = {"mode": mode, "calls": {}, "shapes": {}}


def count(name):
    report["calls"][name] = report["calls"].get(name, 0) + 1


def shapes_of(x):
    if isinstance(x, (list, tuple)):
        return [shapes_of(t) for t in x]
    return None if x is None else list(x.shape) if hasattr(x, "shape") else type(x).__name__


from vid2vid_amd.models import models as amd_models
_rank_forward = amd_models.RankModel.forward


def rank_forward(self, *a, **k):
    out = _rank_forward(self, *a, **k)
    name = type(self.module).__name__
    count(name + ".forward")
    report["shapes"].setdefault(name, shapes_of(out))
    if name == "Vid2VidModelD":
        # dry run: nothing executed, the loss tensors hold whatever the allocator returned.  The reference's printer skips losses
        # that are exactly 0 (util/visualizer.py print_current_errors), so give them a defined non-zero value -- the test that
        # every loss NAME reaches the log must not depend on heap garbage (it failed under MALLOC_PERTURB_=255, which zeroes mallocs)
        for o in out:
            o.data.fill_(1.0)
    return out


amd_models.RankModel.forward = rank_forward
from vid2vid_amd.models.vid2vid_model_G import Vid2VidModelG
_inference = Vid2VidModelG.inference


def inference(self, *a, **k):
    out = _inference(self, *a, **k)
    count("Vid2VidModelG.inference")
    report["shapes"].setdefault("inference", shapes_of(out))
    return out


Vid2VidModelG.inference = inference
from vid2vid_amd import optim
_step = optim.FusedAdam.step


def step(self, *a, **k):
    count("optimizer.step")
    return _step(self, *a, **k)


optim.FusedAdam.step = step

# ---- synthetic data in place of data/data_loader.py (folder walking / PIL decoding are out of scope) ----
from data.base_dataset import BaseDataset          # the reference's own tensor contract


class SyntheticDataset(BaseDataset):
    """Tensors shaped like TemporalDataset / TestDataset items (data/temporal_dataset.py:31-75, data/test_dataset.py:26-60)."""

    def initialize(self, opt):
        self.opt = opt
        self.n_seqs = 2
        self.H, self.W = opt.loadSize // 2, opt.loadSize
        self.n_frames_total = opt.n_frames_total if opt.isTrain else 1
        self.seq_len_max = 30
        self.test_frames = 4

    def __len__(self):
        return self.n_seqs if self.opt.isTrain else self.n_seqs * self.test_frames

    def __getitem__(self, index):
        opt = self.opt
        g = torch.Generator().manual_seed(index)
        if opt.isTrain:
            T = self.n_frames_total + opt.n_frames_G - 1
            change_seq = False
        else:
            T = opt.n_frames_G
            change_seq = index % self.test_frames == 0 and index > 0
        A = torch.randint(0, opt.label_nc, (T, self.H, self.W), generator=g).float()          # label ids as floats
        B = torch.rand(T * opt.output_nc, self.H, self.W, generator=g) * 2 - 1
        inst = torch.randint(0, 20, (T, self.H, self.W), generator=g).float()
        return {"A": A, "B": B, "inst": inst, "A_path": "seq%02d/frame%04d.png" % (index // 4, index % 4), "change_seq": change_seq}


class SyntheticLoader:
    def __init__(self, opt):
        self.dataset = SyntheticDataset()
        self.dataset.initialize(opt)
        self.loader = torch.utils.data.DataLoader(self.dataset, batch_size=opt.batchSize, shuffle=False, num_workers=0)

    def load_data(self):
        return self.loader

    def __len__(self):
        return len(self.dataset)


dl = types.ModuleType("data.data_loader")
dl.CreateDataLoader = lambda opt: SyntheticLoader(opt)
sys.modules["data.data_loader"] = dl

# weights that the reference downloads (FlowNet2, first-frame nets) are random-init here: the flag lives on `opt`
import options.base_options as bo
_parse = bo.BaseOptions.parse


def parse(self, *a, **k):
    opt = _parse(self, *a, **k)
    opt.random_init_ok = True
    opt.precision = "fp32"
    return opt


bo.BaseOptions.parse = parse

common = ["--name", "dropin", "--checkpoints_dir", os.path.join(work, "ckpt"), "--label_nc", "35", "--loadSize", "128",
          "--use_instance", "--fg", "--ngf", "8", "--n_blocks", "2", "--n_blocks_local", "1", "--n_downsample_G", "2",
          "--n_scales_spatial", "2", "--gpu_ids", "-1", "--n_gpus_gen", "1", "--nThreads", "0"]
if mode == "train":
    sys.argv = ["train.py"] + common + ["--ndf", "8", "--num_D", "2", "--n_scales_temporal", "2", "--n_frames_total", "8",
                                         "--max_frames_per_gpu", "2", "--niter", "2", "--niter_decay", "0", "--niter_step", "1",
                                         "--niter_fix_global", "1", "--no_vgg", "--print_freq", "1", "--display_freq", "100000", "--no_html"]
    runpy.run_path(os.path.join(REF, "train.py"), run_name="__main__")
else:
    sys.argv = ["test.py"] + common + ["--use_real_img", "--how_many", "6", "--results_dir", os.path.join(work, "results")]
    runpy.run_path(os.path.join(REF, "test.py"), run_name="__main__")
    report["saved"] = sorted(f for _, _, fs in os.walk(os.path.join(work, "results")) for f in fs)[:12]
report["checkpoints"] = sorted(f for _, _, fs in os.walk(os.path.join(work, "ckpt")) for f in fs)
print("DROPIN_REPORT " + json.dumps(report))
