#!/usr/bin/env python3
"""Host-side cost of one training chunk: bench.py's --mode train step in DRY-RUN mode on the CPU (every launch is argument-
checked by the library and returns at once), under cProfile.  The training step is host-bound (removing 8 ms of kernels
did not move the wall time, profiles/r02_a15_*): this profile shows where the Python / ctypes time per launch goes.

    python scripts/host_profile_train.py [steps] > profiles/rNN_host_profile_train.txt
"""
import cProfile
import io
import os
import pstats
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vid2vid_amd import networks as N
N.set_record_only(True)
from vid2vid_amd import synthetic
from vid2vid_amd.options import make_opt
from vid2vid_amd.models import create_model
from vid2vid_amd.models.models import create_optimizer
from vid2vid_amd.lib import lib

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
H, W = int(os.environ.get("HP_H", 256)), int(os.environ.get("HP_W", 512))
ngf = int(os.environ.get("HP_NGF", 128))
opt = make_opt(isTrain=True, label_nc=35, use_instance=True, fg=True, random_init_ok=True, loadSize=W, precision="bf16",
               gpu_ids=[], n_gpus_gen=1, n_scales_spatial=1, num_D=2, n_frames_total=6, max_frames_per_gpu=2, niter_fix_global=0, ngf=ngf)
_so = sys.stdout
sys.stdout = sys.stderr
models = create_model(opt)
modelG, modelD, flowNet, optimizer_G, optimizer_D, optimizer_D_T = create_optimizer(opt, models)
sys.stdout = _so
tG, tD, t_scales = opt.n_frames_G, opt.n_frames_D, opt.n_scales_temporal
n_frames_total, n_frames_load = opt.n_frames_total, modelG.module.n_frames_load
n_seq = n_frames_total + tG - 1
lab, inst, frames = synthetic.label2city_sequence(n_seq, H, W, seed=1, device="cpu")
A_all, I_all, B_all = lab.view(1, n_seq, 1, H, W), inst.view(1, n_seq, 1, H, W), frames
state = {"i": 0, "prev": None, "frames_all": (None, None, None, None)}


def reshape(ts):
    return [None if t is None else t.contiguous().view(-1, t.size(2), t.size(3), t.size(4)) for t in ts]


def loss_backward(loss, optimizer):
    optimizer.zero_grad()
    loss.backward()
    optimizer.step()


def step():
    i = state["i"]
    if i == 0:
        state["prev"], state["frames_all"] = None, (None, None, None, None)
    te = i + n_frames_load + tG - 1
    a, b, ins = A_all[:, i:te], B_all[:, i:te], I_all[:, i:te]
    fake_B, fake_B_raw, flow, weight, real_A, real_Bp, fake_B_last = modelG(a, b, ins, state["prev"])
    real_B_prev, real_B = real_Bp[:, :-1], real_Bp[:, 1:]
    flow_ref, conf_ref = flowNet(real_B, real_B_prev)
    fake_B_prev = modelG.module.compute_fake_B_prev(real_B_prev, state["prev"], fake_B)
    state["prev"] = fake_B_last
    losses = modelD(0, reshape([real_B, fake_B, fake_B_raw, real_A, real_B_prev, fake_B_prev, flow, weight, flow_ref, conf_ref]))
    loss_dict = dict(zip(modelD.module.loss_names, [torch.mean(x) for x in losses]))
    state["frames_all"], skipped = modelD.module.get_all_skipped_frames(state["frames_all"], real_B, fake_B, flow_ref, conf_ref,
                                                                         t_scales, tD, n_frames_load, i, flowNet)
    loss_dict_T = []
    for s in range(t_scales):
        if skipped[0][s] is not None:
            lt = modelD(s + 1, [f[s] for f in skipped])
            loss_dict_T.append(dict(zip(modelD.module.loss_names_T, [torch.mean(x) for x in lt])))
    loss_G, loss_D, loss_D_T, t_act = modelD.module.get_losses(loss_dict, loss_dict_T, t_scales)
    loss_backward(loss_G, optimizer_G)
    loss_backward(loss_D, optimizer_D)
    for s in range(t_act):
        loss_backward(loss_D_T[s], optimizer_D_T[s])
    state["i"] = i + n_frames_load if (i // n_frames_load + 1) < max(n_frames_total // n_frames_load, 1) else 0


for _ in range(3):
    step()                     # one sequence: all temporal scales active by the last chunk
import gc
gc.collect(); gc.freeze()
t0 = time.perf_counter()
for _ in range(steps):
    step()
print("# dry-run host time per chunk (no profiler): %.1f ms, %dx%d, ngf %d" % ((time.perf_counter() - t0) / steps * 1e3, W, H, ngf))
t0 = time.perf_counter()
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    step()
pr.disable()
dt = (time.perf_counter() - t0) / steps
print("# dry-run host time per chunk (under cProfile): %.1f ms, %dx%d, ngf %d" % (dt * 1e3, W, H, ngf))
for key in ("tottime", "cumulative"):
    sio = io.StringIO()
    pstats.Stats(pr, stream=sio).strip_dirs().sort_stats(key).print_stats(45)
    print(sio.getvalue()[:9000])
