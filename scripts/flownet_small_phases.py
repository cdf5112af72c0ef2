#!/usr/bin/env python3
"""FlowNet2's small layers (below 1/16 resolution, ~25 us per launch whatever their FLOP: profiles/r06_v37_flownet2_per_launch.txt) on the
generic tiles, phase by phase (needs a V2V_STAMP_MASK build: since round 6 its results are right, conv_igemm_kernel.h).
Where a conv launch's time goes, per workgroup: constant-rate wall-clock stamps written by the kernels themselves
(v2v_conv_debug_clocks: 0 entry, 1 first loads issued, 2 first tile landed, 3 main loop done, 4 outputs stored, 5 statistics row
published, 6 exit).  For every shape: the launch's span (first entry -> last exit), and per phase the median / max over workgroups.
    python scripts/kernel_phases.py > gpurun_out/kernel_phases.txt"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
import torch.nn as nn
from vid2vid_amd import lib as L
from vid2vid_amd.lib import lib
from vid2vid_amd.engine import Engine

eng = Engine("cuda:0", L.BF16)
NWG = 1 << 14
buf = torch.zeros(NWG * 8, dtype=torch.int64, device="cuda:0")
PH = ["entry->issued", "issued->landed", "main loop", "stores", "stats row", "finalize/exit"]


def report(name, run, reps=5):
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    rows = []
    for _ in range(reps):
        buf.zero_()
        torch.cuda.synchronize()
        lib.v2v_conv_debug_clocks(C.c_void_p(buf.data_ptr()))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record(); e1.synchronize()
        lib.v2v_conv_debug_clocks(None)
        t = buf.view(NWG, 8).cpu().double() * 0.01            # 100 MHz ticks -> us
        t = t[t[:, 0] > 0]
        rows.append((t, e0.elapsed_time(e1) * 1e3))
    t, ev = sorted(rows, key=lambda r: r[1])[len(rows) // 2]
    t0 = t[:, 0].min()
    last = t[:, 6].clone()
    last[last == 0] = t[:, 4][last == 0]
    span = (last.max() - t0).item()
    line = "%-34s wgs %4d  event %6.1f us  span %6.1f us  entry spread %5.1f us |" % (name, t.shape[0], ev, span, (t[:, 0].max() - t0).item())
    for k, ph in enumerate(PH):
        a, b = t[:, k], t[:, k + 1]
        ok = (a > 0) & (b > 0)
        if ok.any():
            d = (b - a)[ok]
            line += " %s med %5.1f max %5.1f |" % (ph, d.median().item(), d.max().item())
    print(line, flush=True)



# (name, cin, cout, k, stride, H, W, N, [(tile, split-K)])
SHAPES = [("conv6_1 1024->1024 k3 @8x4 x3", 1024, 1024, 3, 1, 4, 8, 3, [(9, 6), (9, 8), (9, 1), (3, 8), (10, 8)]),
          ("conv6 512->1024 k3/s2 @16x8 x3", 512, 1024, 3, 2, 8, 16, 3, [(9, 6), (9, 8), (9, 1)]),
          ("conv5_1 512->512 k3 @16x8 x3", 512, 512, 3, 1, 8, 16, 3, [(9, 4), (9, 8), (9, 1)]),
          ("predict_flow6 1024->2 k3 @8x4 x3", 1024, 2, 3, 1, 4, 8, 3, [(4, 8), (4, 1), (9, 8)]),
          ("conv4_1 512->512 k3 @32x16 x3", 512, 512, 3, 1, 16, 32, 3, [(9, 2), (9, 4), (3, 2)])]
with torch.no_grad():
    for name, cin, cout, k, st, H, W, N, tiles in SHAPES:
        mod = nn.Conv2d(cin, cout, k, stride=st, padding=k // 2).to("cuda:0")
        x = eng.pack(torch.randn(N, cin, H, W, device="cuda:0"))
        for cfg in tiles:
            eng.tile_override[(cin, cout, k, st, 0)] = (cfg[0], cfg[1], 0)
            try:
                report("%s t%d/S%d" % (name, cfg[0], cfg[1]),
                       lambda: eng.conv(x, mod, L.PAD_ZERO, None, L.OUT_ACT_NHWC, L.ACT_LEAKY, 0.1, 1.0, label="flownet"))
            except Exception as ex:
                print("%s t%d/S%d: %r" % (name, cfg[0], cfg[1], ex))
