#!/usr/bin/env python3
"""Where the per-frame time outside the frame graph goes: Vid2VidModelG.inference() = stage inputs (2 D2D copies) + one
hipGraph launch + 2 output clones.  Times N frames of: the full call, the graph launches alone, and the call with the copies /
clones removed one at a time (monkey-patched; results of those variants are not used)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vid2vid_amd import synthetic
from vid2vid_amd.options import make_opt
from vid2vid_amd.models import create_model
dev = torch.device("cuda", 0)
H, W = 256, 512
import shutil, tempfile
tmp = os.path.join(tempfile.gettempdir(), "fo_tune.json")
shutil.copyfile(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "tune_cache.json"), tmp)
os.environ["V2V_TUNE_CACHE"] = tmp
opt = make_opt(label_nc=35, use_instance=True, fg=True, use_real_img=True, random_init_ok=True, loadSize=W, precision="bf16", gpu_ids=[0])
so = sys.stdout; sys.stdout = sys.stderr
model = create_model(opt)
sys.stdout = so
tG, L = 3, 16
lab, inst, frames = synthetic.label2city_sequence(L + tG, H, W, seed=1234, device=dev)
A, I = lab.view(1, L + tG, 1, H, W), inst.view(1, L + tG, 1, H, W)
model.fake_B_prev = None
model.inference(A[:, 0:tG], frames[:, :tG - 1], I[:, 0:tG])
fp = model._active_plan
N = 60


def timed(fn):
    for t in range(5):
        fn(t + 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(N):
        fn(t + 6)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / N * 1e3


def full(t):
    k = t % L
    return model.inference(A[:, k:k + tG], None, I[:, k:k + tG])


def graph_only(t):
    fp.run()


def copies_graph(t):
    k = t % L
    fp.labels.copy_(A[0, k:k + tG, 0]); fp.inst.copy_(I[0, k:k + tG, 0])
    fp.run()


def graph_clones(t):
    fp.run()
    return fp.out["fake_B"].clone(), fp.out["real_A_last"].clone()


def graph_clone_small(t):
    fp.run()
    return fp.out["fake_B"].clone()


for name, fn in (("inference() (stage + graph + clones)", full), ("graph launches only", graph_only), ("stage + graph", copies_graph),
                 ("graph + both clones", graph_clones), ("graph + fake_B clone only", graph_clone_small), ("inference() again", full)):
    print("%-40s %.4f ms/frame" % (name, timed(fn)))
