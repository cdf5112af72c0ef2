#!/usr/bin/env python3
"""Where does the bf16 path's error come from?    CPU study on the oracle: the bf16 engine rounds
(a) packed weights and (b) every STORED activation (the output of norm + activation, the residual sums) to bf16 and
accumulates in fp32 -- emulated here by rounding exactly those tensors inside oracle/vid2vid_oracle.py's layer walker, one
layer group at a time, on the benchmark's own weights and inputs (label2city WxH, reference initialisers, seed 0, flow head
x 0.1).  Prints per-head max / mean relative error (the measure of bench.py / tests/util.py) of each policy against the
all-fp32 oracle.  Test / analysis infrastructure only.

    python scripts/bf16_attribution.py [W H]
"""
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import vid2vid_oracle as O          # noqa: E402

GROUPS = ("stem", "down", "res", "up", "head")
POLICY = {}          # group -> True: the group runs in bf16 (weights + stored outputs rounded)
ROUND_W, ROUND_A = [True], [True]   # which of the two roundings are applied (weights / stored activations)
STREAM_FP32 = [False]  # ResnetBlock: keep the residual stream (x + h) in fp32, round only the conv inputs
CUR = [None]
BRANCH = [""]        # prefix of the Sequential being walked (model_res_flow, indv_up, ...)
BRANCH_FP32 = [()]   # prefixes forced to fp32 whatever the group policy says


def r16(t):
    return t.bfloat16().float()


def rw(t):
    return r16(t) if ROUND_W[0] else t


def ra(t):
    return r16(t) if ROUND_A[0] else t


def on():
    return POLICY.get(CUR[0], False) and not any(BRANCH[0].startswith(p) for p in BRANCH_FP32[0])


_conv0, _convT0 = O._conv, O._convT


def _conv(sd, key, x, stride=1, padding=0):
    if on():
        return F.conv2d(ra(x), rw(sd[key + ".weight"]), sd.get(key + ".bias"), stride=stride, padding=padding)
    return _conv0(sd, key, x, stride, padding)


def _convT(sd, key, x):
    if on():
        return F.conv_transpose2d(ra(x), rw(sd[key + ".weight"]), sd.get(key + ".bias"), stride=2, padding=1, output_padding=1)
    return _convT0(sd, key, x)


O._conv, O._convT = _conv, _convT
W0 = O._Walker


class Walker(W0):
    def _run(self, group, fn, x, *a):
        CUR[0], BRANCH[0] = group, self.p
        y = fn(self, x, *a)
        if on() and group != "head":                       # heads write fp32 NCHW
            y = ra(y)
        CUR[0] = None
        return y

    def stem7(self, x, act=True): return self._run("stem", W0.stem7, x, act)
    def down3(self, x): return self._run("down", W0.down3, x)
    def up3(self, x): return self._run("up", W0.up3, x)
    def head7(self, x, act=None): return self._run("head", W0.head7, x, act)

    def resblock(self, x):
        CUR[0], BRANCH[0] = "res", self.p
        if not on():
            y = W0.resblock(self, x)
            CUR[0] = None
            return y
        k = "%s.%d.conv_block" % (self.p, self.i)
        h = ra(F.relu(O._norm(self.sd, k + ".2", O._conv(self.sd, k + ".1", O._reflect(x, 1)), self.norm)))
        h = O._norm(self.sd, k + ".6", O._conv(self.sd, k + ".5", O._reflect(h, 1)), self.norm)
        self.i += 1
        y = x + h
        if not STREAM_FP32[0]:
            y = ra(y)
        CUR[0] = None
        return y


O._Walker = Walker


def main():
    W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (512, 256)
    from vid2vid_amd import networks as N, synthetic
    from vid2vid_amd.options import make_opt
    torch.manual_seed(0)
    opt = make_opt(label_nc=35, use_instance=True, fg=True, use_real_img=True, random_init_ok=True, loadSize=W)
    net = N.define_G(108, 3, 6, opt.ngf, "composite", 3, "batch", 0, [], opt)
    with torch.no_grad():
        net.model_final_flow[1].weight.mul_(0.1)
    sd = {k: v.detach().float() for k, v in net.state_dict().items()}
    lab, inst, frames = synthetic.label2city_sequence(4, H, W, seed=1234)
    orc = lambda: O.InferenceOracle([sd], 35, True, True, [26], 3, 9, 3)

    def run(policy, stream_fp32=False, branch_fp32=(), w=True, a=True):
        POLICY.clear(); POLICY.update(policy); STREAM_FP32[0] = stream_fp32; BRANCH_FP32[0] = tuple(branch_fp32)
        ROUND_W[0], ROUND_A[0] = w, a
        o = orc()
        with torch.no_grad():
            fake, _ = o.step(lab[0:3].view(1, 3, 1, H, W), frames[:, :2], inst[0:3].view(1, 3, 1, H, W))
        return dict(fake_B=fake, raw=o.last["raw0"], flow=o.last["flow0"], weight=o.last["weight0"])

    t0 = time.time()
    ref = run({})
    print("# %dx%d, one frame from the given real frames; fp32 oracle %.1f s; columns: max_rel / mean_rel" % (W, H, time.time() - t0))
    allb = {g: True for g in GROUPS}
    flow_br = ("model_res_flow", "model_up_flow", "model_final_flow", "model_final_w")
    cases = [("all bf16", allb, False, ()),
             ("all bf16, residual stream fp32", allb, True, ()),
             ("only stem bf16", {"stem": True}, False, ()), ("only down bf16", {"down": True}, False, ()),
             ("only res bf16", {"res": True}, False, ()), ("only res bf16, stream fp32", {"res": True}, True, ()),
             ("only up bf16", {"up": True}, False, ()),
             ("only head bf16", {"head": True}, False, ()),
             ("all bf16 but heads", dict(allb, head=False), False, ()),
             ("all bf16 but up + heads", dict(allb, head=False, up=False), False, ()),
             ("all bf16 but up + heads, stream fp32", dict(allb, head=False, up=False), True, ()),
             ("all bf16 but the flow branch", allb, False, flow_br),
             ("all bf16 but the flow branch, stream fp32", allb, True, flow_br),
             ("all bf16 but stem + down", dict(allb, stem=False, down=False), False, ())]
    cases = [c + (True, True) for c in cases]
    cases += [("all groups: weights rounded only", allb, False, (), True, False),
              ("all groups: stored activations rounded only", allb, False, (), False, True),
              ("stem + down: weights only", {"stem": True, "down": True}, False, (), True, False),
              ("stem + down: activations only", {"stem": True, "down": True}, False, (), False, True),
              ("res: weights only", {"res": True}, False, (), True, False),
              ("res: activations only", {"res": True}, False, (), False, True)]
    for name, pol, sfp, bfp, w_, a_ in cases:
        got = run(pol, sfp, bfp, w_, a_)
        cols = []
        for k in ("fake_B", "raw", "flow", "weight"):
            r, g = ref[k], got[k]
            e = (g - r).abs() / (r.abs() + r.pow(2).mean().sqrt().item() + 1e-12)
            cols.append("%s %.2e / %.2e" % (k, e.max().item(), e.mean().item()))
        print("%-44s %s" % (name, "   ".join(cols)), flush=True)


if __name__ == "__main__":
    main()
