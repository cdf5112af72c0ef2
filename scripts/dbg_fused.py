import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn
from vid2vid_amd import lib as L
from vid2vid_amd.engine import Engine
DEV = "cuda:0"
for prec in ("fp32", "bf16"):
    eng = Engine(DEV, L.F32 if prec == "fp32" else L.BF16)
    torch.manual_seed(1)
    cin, cout, H, W, N = 128, 72, 24, 64, 1
    with torch.no_grad():
        convs = [nn.Conv2d(cin, cout, 3, padding=0).to(DEV) for _ in range(2)]
        nf = [nn.BatchNorm2d(cout).to(DEV) for _ in range(2)]
        nu = [nn.BatchNorm2d(cout).to(DEV) for _ in range(2)]
        xa = [eng.pack(torch.randn(N, cin, H, W).to(DEV)) for i in range(2)]
        ra = [eng.pack(torch.randn(N, cout, H, W).to(DEV)) for _ in range(2)]
        eng.pair_override = (80, 1)
        out = {}
        for fused, norms in ((True, nf), (False, nu)):
            eng.fused_norm = fused
            y = eng.conv_group_pair(xa[0], convs[0], norms[0], xa[1], convs[1], norms[1], L.PAD_REFLECT, 1, L.ACT_RELU, 0.0,
                                    adds_a=(ra[0], None), adds_b=(ra[1], None), labels=("a", "b"))
            ss = []
            for sset in (0, 1):
                with eng.scratch_set(sset):
                    ss.append(eng.scratch("scale_shift", 4 * cout)[:4 * cout].clone().view(4, cout))
                    rows = 6
                    st = eng.scratch("stats", rows * cout * 2)[:rows * cout * 2].clone().view(rows, cout, 2)
            out[fused] = (y, ss, st)
            print(prec, "fused" if fused else "unfused", eng.conv_log[-1].get("fused_norm"), "stats row0 ch0..3", st[0, :4].tolist())
        for k in range(2):
            d = (out[True][1][k] - out[False][1][k]).abs().max(1).values
            print(prec, "member", k, "ss max abs diff (scale, shift, mean, invstd):", d.tolist(),
                  " y diff", (out[True][0][k].t.float() - out[False][0][k].t.float()).abs().max().item())
        print(prec, "stats rows diff", (out[True][2] - out[False][2]).abs().max().item())
