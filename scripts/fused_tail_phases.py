#!/usr/bin/env python3
"""Round 6: where the fused-norm tail of the dominant launch goes.  The PAIR of 1024 -> 1024 3x3 ResnetBlock
convolutions with norm + residual fused (tile 90, 256 workgroups) on a V2V_STAMP_MASK=0xff build: every workgroup's thread 0 stamps the
100 MHz device clock at 0 entry, 1 first loads issued, 2 first slices landed, 3 K loop done, 4 statistics row in memory + arrival ticket
issued, 5 every workgroup of its channel tile has arrived, 6 scale / shift in LDS, 7 normalised tile stored.  Cold weights (memset
between launches).  Prints per phase the median / max over workgroups, and the phases of the workgroup that exits LAST.

    V2V_LIB_PATH=<stamp build> python scripts/fused_tail_phases.py
"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn
from vid2vid_amd import lib as L
from vid2vid_amd.lib import lib
from vid2vid_amd.engine import Engine

DEV = "cuda:0"
eng = Engine(DEV, L.BF16)
THRASH = torch.empty(96 << 20, dtype=torch.float32, device=DEV)
torch.manual_seed(0)
cin = cout = 1024
H, W = 32, 64
convs = [nn.Conv2d(cin, cout, 3).to(DEV) for _ in range(2)]
norms = [nn.BatchNorm2d(cout).to(DEV) for _ in range(2)]
xs = [eng.pack(torch.randn(1, cin, H, W, device=DEV)) for _ in range(2)]
res = [eng.pack(torch.randn(1, cout, H, W, device=DEV)) for _ in range(2)]
ya, yb = eng.empty_act(1, H, W, cout), eng.empty_act(1, H, W, cout)


def run():
    eng.pair_override = (90, 1)
    ssa = eng.scratch("scale_shift", 4 * cout)
    with eng.scratch_set(1):
        ssb = eng.scratch("scale_shift", 4 * cout)
    eng.conv_pair(xs[0], convs[0], xs[1], convs[1], L.PAD_REFLECT, 1, ((norms[0], ssa), (norms[1], ssb)), ("a", "b"),
                  fuse=(L.ACT_NONE, 0.0, (res[0], None), (res[1], None), ya, yb))


NWG = 1 << 12
buf = torch.zeros(NWG * 8, dtype=torch.int64, device=DEV)
PH = ["entry->issued", "issued->landed", "K loop", "stats row + ack + ticket", "wait for the channel tile", "rows -> scale/shift", "normalise + store"]
for _ in range(3):
    run()
torch.cuda.synchronize()
rows = []
for _ in range(9):
    buf.zero_()
    THRASH.zero_()
    lib.v2v_conv_debug_clocks(C.c_void_p(buf.data_ptr()))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); e1.synchronize()
    lib.v2v_conv_debug_clocks(None)
    t = buf.view(NWG, 8).cpu().double() * 0.01
    t = t[t[:, 0] > 0]
    rows.append((t, e0.elapsed_time(e1) * 1e3))
t, ev = sorted(rows, key=lambda r: r[1])[len(rows) // 2]
t0 = t[:, 0].min()
print("pair of 1024->1024 3x3 @64x32, tile 90, fused norm + residual: %d workgroups stamped, HIP-event time %.1f us, device span (first entry -> last stamp 7) %.1f us, entry spread %.1f us"
      % (t.shape[0], ev, (t[:, 7].max() - t0).item(), (t[:, 0].max() - t0).item()))
for k, ph in enumerate(PH):
    d = t[:, k + 1] - t[:, k]
    print("  %-28s median %6.2f  min %6.2f  max %6.2f us" % (ph, d.median().item(), d.min().item(), d.max().item()))
for k in (3, 4, 5, 6, 7):
    v = t[:, k] - t0
    print("  stamp %d reached at (since first entry): min %6.2f  median %6.2f  max %6.2f us" % (k, v.min().item(), v.median().item(), v.max().item()))
i = int(t[:, 7].argmax())
print("  the workgroup that exits last: " + "  ".join("%s %.2f" % (PH[k], (t[i, k + 1] - t[i, k]).item()) for k in range(7)))
j = int(t[:, 3].argmax())
print("  the workgroup whose K loop ends last (%.2f us after the first entry): " % (t[j, 3] - t0).item()
      + "  ".join("%s %.2f" % (PH[k], (t[j, k + 1] - t[j, k]).item()) for k in range(3, 7)))
print("  tail behind the slowest K loop: %.2f us" % (t[:, 7].max() - t[:, 3].max()).item())
