#!/usr/bin/env python3
"""Per-launch table of one FlowNet2 pass (models/flownet.py plan, bf16): every recorded op ALONE on the chip (v2v_plan_profile),
the convolutions joined with their algorithmic FLOP, and the hipGraph replay time of the whole pass beside the sum.
    python scripts/flownet2_profile.py [B=3] [H=256] [W=512] > profiles/rNN_flownet2_per_launch.txt"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vid2vid_amd.options import make_opt
from vid2vid_amd.models.flownet import FlowNet

B = int(sys.argv[1]) if len(sys.argv) > 1 else 3
H = int(sys.argv[2]) if len(sys.argv) > 2 else 256
W = int(sys.argv[3]) if len(sys.argv) > 3 else 512
_so = sys.stdout
sys.stdout = sys.stderr
torch.manual_seed(0)
opt = make_opt(isTrain=True, no_vgg=True, precision="bf16", gpu_ids=[0], random_init_ok=True)
fn = FlowNet(); fn.initialize(opt)
eng = fn.engine
im1, im2 = torch.rand(B, 3, H, W, device="cuda:0"), torch.rand(B, 3, H, W, device="cuda:0")
n0 = len(eng.conv_log)
with torch.no_grad():                                               # as FlowNet.forward runs it (the tile search needs grad mode off)
    fn.compute_flow_and_conf(im1, im2)
torch.cuda.synchronize()
fp = next(iter(fn._plans.values()))
convs = eng.conv_log[len(eng.conv_log) - fp.n_convs:]
sys.stdout = _so
reps = 30
for _ in range(5):
    fp.plan.launch()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    fp.plan.launch()
torch.cuda.synchronize()
graph_ms = (time.perf_counter() - t0) / reps * 1e3
prof = fp.plan.profile()
prof2 = fp.plan.profile()
prof = [(a[0], a[1], min(a[2], b[2])) for a, b in zip(prof, prof2)]
tot = sum(p[2] for p in prof)
print("# FlowNet2 bf16, %d pair(s) at %dx%d: %d launches, hipGraph replay %.3f ms (%.1f pairs/s), sum of the launches alone %.3f ms; %.1f GFLOP per pair"
      % (B, W, H, len(prof), graph_ms, B / graph_ms * 1e3, tot, fp.conv_flops / B / 1e9))
print("# whole pass: %.1f TFLOP/s = %.2f %% of the 2.5 PFLOP/s bf16 peak" % (fp.conv_flops / graph_ms / 1e9, fp.conv_flops / graph_ms / 1e9 / 25.0))
ci = 0
rows = []
for i, (op, label, ms) in enumerate(prof):
    c = None
    if op.startswith("conv") and ci < len(convs):
        c = convs[ci]; ci += 1
    rows.append((ms, i, op, label, c))
print("%4s %-22s %-28s %9s %9s %8s  %s" % ("#", "op", "cin->cout k/s @WxH (N)", "us", "GFLOP", "TFLOP/s", "tile,S"))
by_op = {}
for ms, i, op, label, c in rows:
    by_op.setdefault(op, [0, 0.0]); by_op[op][0] += 1; by_op[op][1] += ms
for ms, i, op, label, c in sorted(rows, key=lambda r: -r[0])[:70]:
    if c is not None:
        shape = "%d->%d k%d%s/s%d @%dx%d (%d)" % (c["cin"], c["cout"], c["KH"], "T" if c.get("transposed") else "", c.get("stride", 1), c.get("W", 0), c.get("H", 0), c["N"])
        print("%4d %-22s %-28s %9.1f %9.2f %8.1f  %s,%s" % (i, op[:22], shape, ms * 1e3, c["flops"] / 1e9, c["flops"] / ms / 1e9, c.get("tile"), c.get("splitk")))
    else:
        print("%4d %-22s %-28s %9.1f" % (i, op[:22], label[:28], ms * 1e3))
print()
for op, (n, ms) in sorted(by_op.items(), key=lambda kv: -kv[1][1]):
    print("# %-28s %4d launches %9.1f us" % (op, n, ms * 1e3))
