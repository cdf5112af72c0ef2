#!/usr/bin/env python3
"""Round-5 experiments on the dominant launch of the 512x256 frame: the PAIR of 1024 -> 1024 3x3 ResnetBlock convolutions with
norm + ReLU / residual fused (tile 90 today), alone on the chip, cold weights (384 MB memset between launches) and warm,
interleaved rounds, HIP events.  Variants:
    90  the shipped tile (K pairs, 5 weight stages, one barrier per step)
    98  (ii) the weight ring one stage deeper (6 stages)
    97  one barrier per TWO steps (6 stages)            99  the same with 7 stages + static priority for waves 4-7
    82 / 80 / 91 ... other shipped tiles for reference
and (i) the shape a two-sequence grouped launch would have: M = 4096 pixels (N = 2) on the 256 px x 128 tile 81 -- raw output +
statistics only (a fused norm over N = 2 would pool the statistics of the two sequences, which the reference does not do).
Every variant's outputs are compared with tile 90's bit for bit.

    python scripts/dom_bench.py > gpurun_out/dom_bench.txt
"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn
from vid2vid_amd import lib as L
from vid2vid_amd.engine import Engine

DEV = "cuda:0"
eng = Engine(DEV, L.BF16)
THRASH = torch.empty(96 << 20, dtype=torch.float32, device=DEV)
ROUNDS = int(os.environ.get("DOM_ROUNDS", "9"))
TILES = [int(t) for t in os.environ.get("DOM_TILES", "90,82,91").split(",")]


def timed(fn, cold):
    if cold:
        THRASH.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3


def bench(variants, cold):
    for fn in variants.values():
        for _ in range(2):
            fn()
    torch.cuda.synchronize()
    ts = {k: [] for k in variants}
    for _ in range(ROUNDS):
        for k, fn in variants.items():
            ts[k].append(timed(fn, cold))
    return {k: (sorted(v)[len(v) // 2], min(v)) for k, v in ts.items()}


torch.manual_seed(0)
cin = cout = 1024
H, W = 32, 64
convs = [nn.Conv2d(cin, cout, 3).to(DEV) for _ in range(2)]
norms = [nn.BatchNorm2d(cout).to(DEV) for _ in range(2)]
with torch.no_grad():
    for n_ in norms:
        n_.weight.normal_(1, 0.02)
xs = [eng.pack(torch.randn(1, cin, H, W, device=DEV)) for _ in range(2)]
res = [eng.pack(torch.randn(1, cout, H, W, device=DEV)) for _ in range(2)]
flops = 2.0 * H * W * cout * cin * 9
outs = {}


def pair_fused(tile):
    ya, yb = eng.empty_act(1, H, W, cout), eng.empty_act(1, H, W, cout)

    def fn():
        eng.pair_override = (tile, 1)
        ssa = eng.scratch("scale_shift", 4 * cout)
        with eng.scratch_set(1):
            ssb = eng.scratch("scale_shift", 4 * cout)
        # second convolution of a ResnetBlock: norm, no activation, + the block input (networks.py:591-593)
        eng.conv_pair(xs[0], convs[0], xs[1], convs[1], L.PAD_REFLECT, 1, ((norms[0], ssa), (norms[1], ssb)), ("a", "b"),
                      fuse=(L.ACT_NONE, 0.0, (res[0], None), (res[1], None), ya, yb))
    outs[tile] = (ya, yb)
    return fn


variants = {}
for t in TILES:
    try:
        f = pair_fused(t)
        f()
        torch.cuda.synchronize()
        variants["pair t%d fused" % t] = f
    except Exception as ex:                                   # a tile the library refuses is reported, not fatal
        print("tile %d: %s" % (t, str(ex)[:200]))
ref = outs[TILES[0]]
import hashlib
print("checksum of tile %d's two outputs (%s): %s" % (TILES[0], os.path.basename(L.LIB_PATH), hashlib.sha1(b"".join(
    o.t.detach().cpu().contiguous().view(torch.uint8).numpy().tobytes() for o in ref)).hexdigest()[:16]))
for t in TILES[1:]:
    if "pair t%d fused" % t in variants:
        d = max((outs[t][i].t.float() - ref[i].t.float()).abs().max().item() for i in range(2))
        fin = all(torch.isfinite(outs[t][i].t.float()).all().item() for i in range(2))
        print("tile %d vs tile %d: max |diff| %.3e, finite %s" % (t, TILES[0], d, fin))
for cold in (True, False):
    r = bench(variants, cold)
    print("== pair of %d->%d 3x3 @%dx%d bf16, norm + residual fused, %s (%.2f GFLOP per launch), median / min of %d" % (cin, cout, W, H, "cold" if cold else "warm", 2 * flops / 1e9, ROUNDS))
    for k, (us, mn) in r.items():
        print("   %-18s %7.1f us  %6.0f TFLOP/s  frac %.3f   (min %7.1f us, frac %.3f)" % (k, us, 2 * flops / us / 1e6, 2 * flops / us / 1e6 / 2500, mn, 2 * flops / mn / 1e6 / 2500))
    sys.stdout.flush()

# (i) the geometry of a two-sequence grouped launch: N = 2, raw output + statistics, single convolution
x2 = eng.pack(torch.randn(2, cin, H, W, device=DEV))
x1 = xs[0]
key = (cin, cout, 3, 1, 0)
single = {}


def one(x, tile, S):
    def fn():
        eng.tile_override[key] = (tile, S, 0)
        eng.conv(x, convs[0], L.PAD_REFLECT, 1, L.OUT_RAW_F32_NHWC, want_stats=True)
    return fn


for name, x, tile in (("N=1 t90 (128 wgs)", x1, 90), ("N=1 t82 (128 wgs)", x1, 82), ("N=2 t90 (256 wgs)", x2, 90), ("N=2 t81 256x128 (128 wgs)", x2, 81),
                      ("N=2 t85 (4x64 px x 128)", x2, 85)):
    try:
        f = one(x, tile, 1); f(); torch.cuda.synchronize()
        single[name] = (f, x.N)
    except Exception as ex:
        print("%s: %s" % (name, str(ex)[:160]))
for cold in (True, False):
    r = bench({k: v[0] for k, v in single.items()}, cold)
    print("== single 1024->1024 3x3, raw fp32 + statistics, %s" % ("cold" if cold else "warm"))
    for k, (us, mn) in r.items():
        n_ = single[k][1]
        print("   %-28s %7.1f us  %6.0f TFLOP/s  frac %.3f" % (k, us, n_ * flops / us / 1e6, n_ * flops / us / 1e6 / 2500))
    sys.stdout.flush()
