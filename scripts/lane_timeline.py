#!/usr/bin/env python3
"""Concurrent timeline of the 512x256 frame plan (v2v_plan_timeline: every lane on its own stream, timing events around every
op): when each op of each lane starts and ends, per-lane busy time, and the ops on the critical path.
    python scripts/lane_timeline.py > profiles/rNN_lane_timeline.txt"""
import os, sys, shutil, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vid2vid_amd import synthetic
from vid2vid_amd.options import make_opt
from vid2vid_amd.models import create_model
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dev = torch.device("cuda", 0)
H, W = 256, 512
tmp = os.path.join(tempfile.gettempdir(), "lt_tune.json")
shutil.copyfile(os.path.join(ROOT, "profiles", "tune_cache.json"), tmp)
os.environ["V2V_TUNE_CACHE"] = tmp
opt = make_opt(label_nc=35, use_instance=True, fg=True, use_real_img=True, random_init_ok=True, loadSize=W, precision="bf16", gpu_ids=[0])
so = sys.stdout; sys.stdout = sys.stderr
model = create_model(opt)
sys.stdout = so
tG, L = 3, 16
lab, inst, frames = synthetic.label2city_sequence(L + tG, H, W, seed=1234, device=dev)
A, I = lab.view(1, L + tG, 1, H, W), inst.view(1, L + tG, 1, H, W)
model.fake_B_prev = None
for t in range(4):
    model.inference(A[:, t:t + tG], frames[:, :tG - 1] if t == 0 else None, I[:, t:t + tG])
fp = model._active_plan
torch.cuda.synchronize()
runs = [fp.plan.timeline(graph=True) for _ in range(5)]
tl = runs[-1]
end = max(r[4] for r in tl)
print("# frame plan, %d ops, hipGraph replay with a wall-clock stamp kernel around every op: %.3f ms from first to last stamp (5 replays: %s)"
      % (len(tl), end, ", ".join("%.3f" % max(r[4] for r in x) for x in runs)))
busy = {}
for op, label, lane, t0, t1 in tl:
    if op != "lane_wait":
        busy[lane] = busy.get(lane, 0.0) + (t1 - t0)
print("# busy time per lane (sum of op durations as run, kernels of other lanes sharing the chip): " +
      ", ".join("lane %d %.3f ms" % (k, v) for k, v in sorted(busy.items())))
print("# %-5s %9s %9s %8s  %-18s %s" % ("lane", "start us", "end us", "dur us", "op", "label"))
for op, label, lane, t0, t1 in sorted(tl, key=lambda r: r[3]):
    if op == "lane_wait":
        continue
    print("  %-5d %9.1f %9.1f %8.1f  %-18s %s" % (lane, t0 * 1e3, t1 * 1e3, (t1 - t0) * 1e3, op, label))
