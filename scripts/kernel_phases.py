#!/usr/bin/env python3
"""Where a conv launch's time goes, per workgroup: constant-rate wall-clock stamps written by the kernels themselves
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


SHAPES = [("down 128->256 @512x256", 128, 256, 256, 512, 0), ("down 256->512 @256x128", 256, 512, 128, 256, 0),
          ("down 512->1024 @128x64", 512, 1024, 64, 128, 0), ("up 1024->512 @64x32", 1024, 512, 32, 64, 1),
          ("up 512->256 @128x64", 512, 256, 64, 128, 1), ("up 256->128 @256x128", 256, 128, 128, 256, 1)]
TILES = {0: [(18, 1, 0), (14, 1, 0), (15, 1, 0), (17, 1, 0)], 1: [(13, 1, 0), (14, 1, 0), (17, 1, 0)]}
with torch.no_grad():
    for name, cin, cout, H, W, tr in SHAPES:
        mod = (nn.ConvTranspose2d(cin, cout, 3, stride=2, padding=1, output_padding=1) if tr
               else nn.Conv2d(cin, cout, 3, stride=2, padding=1)).to("cuda:0")
        norm = nn.BatchNorm2d(cout).to("cuda:0")
        ss = torch.zeros(4 * cout, device="cuda:0")
        x = eng.pack(torch.randn(1, cin, H, W, device="cuda:0"))
        for cfg in TILES[tr]:
            eng.tile_override[(cin, cout, 3, 2, tr)] = cfg
            for fin in (True, False):
                try:
                    report("%s t%d%s" % (name, cfg[0], "" if fin else " nofin"),
                           lambda: eng.conv(x, mod, L.PAD_ZERO, None, L.OUT_RAW_F32_NHWC, want_stats=True, fin=(norm, ss) if fin else None))
                except Exception as ex:
                    print("%s t%d: %r" % (name, cfg[0], ex))
    # the foreground tower's ResnetBlock convolution (512 -> 512 @64x32) and the 1024-channel one, single launches
    for cin, tiles in ((512, [(91, 2, 0), (91, 1, 0), (83, 1, 0)]), (1024, [(91, 1, 0), (91, 2, 0)])):
        mod = nn.Conv2d(cin, cin, 3, padding=0).to("cuda:0")
        norm = nn.BatchNorm2d(cin).to("cuda:0")
        ss = torch.zeros(4 * cin, device="cuda:0")
        x = eng.pack(torch.randn(1, cin, 32, 64, device="cuda:0"))
        for cfg in tiles:
            eng.tile_override[(cin, cin, 3, 1, 0)] = cfg
            try:
                report("res %d->%d @64x32 t%d/S%d" % (cin, cin, cfg[0], cfg[1]),
                       lambda: eng.conv(x, mod, L.PAD_REFLECT, 1, L.OUT_RAW_F32_NHWC, want_stats=True, fin=(norm, ss)))
            except Exception as ex:
                print("res %d t%d: %r" % (cin, cfg[0], ex))
