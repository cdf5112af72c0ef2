#!/usr/bin/env python3
"""Single-channel-chunk 3x3 layers (64 bf16 input channels: the fine-scale ResnetBlocks): the single-buffer ping-pong tiles 58 / 59
(two workgroups per CU) beside the double-buffer tiles 56 / 57 and the generic / single-phase tiles, cold cache (384 MB memset
between launches), bf16, raw fp32 output + statistics rows as the frame runs them.
    python scripts/one_bench.py > gpurun_out/one_bench.txt"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn
from vid2vid_amd import lib as L
from vid2vid_amd.engine import Engine, _ptr, _stream

eng = Engine("cuda:0", L.BF16)
thrash = torch.empty(96 << 20, dtype=torch.float32, device="cuda:0")


def timed(run, reps=9):
    for _ in range(2):
        run()
    ts = []
    for _ in range(reps):
        thrash.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[len(ts) // 2]


SHAPES = [("G2 res 64->64 @1024x512", 64, 64, 512, 1024), ("G1 fg res 64->64 @512x256", 64, 64, 256, 512),
          ("64->128 @512x256", 64, 128, 256, 512), ("64->64 @256x128", 64, 64, 128, 256),
          ("G2 fg res 32->32 @1024x512", 32, 32, 512, 1024)]       # 64-byte pixels: tiles 140 - 143 through the paired-x view (engine.PairedXConv)
TILES = [(10, 1, 0), (13, 1, 0), (36, 1, 0), (54, 1, 0), (56, 1, 0), (57, 1, 0), (80, 1, 0), (83, 1, 0), (94, 1, 0), (95, 1, 0), (96, 1, 0), (140, 1, 0), (141, 1, 0), (143, 1, 0)]   # (58 / 59: the experiment of profiles/r04_d4_*, not in the tree)
if os.environ.get("ONE_TILES"):
    TILES = [(int(t), 1, 0) for t in os.environ["ONE_TILES"].split(",")]
with torch.no_grad():
    for name, cin, cout, H, W in SHAPES:
        mod = nn.Conv2d(cin, cout, 3).to("cuda:0")
        x = eng.pack(torch.randn(1, cin, H, W, device="cuda:0"))
        out, ref = [], None
        for cfg in TILES:
            eng.tile_override[(cin, cout, 3, 1, 0)] = cfg
            try:
                us = timed(lambda: eng.conv(x, mod, L.PAD_REFLECT, 1, L.OUT_RAW_F32_NHWC, want_stats=True))
            except Exception as e:
                out.append("t%d: n/a" % cfg[0]); continue
            raw = eng.conv(x, mod, L.PAD_REFLECT, 1, L.OUT_RAW_F32_NHWC, want_stats=True)[0][:H * W * cout].clone()
            if ref is None:
                ref = raw
            d = (raw - ref).abs().max().item()
            out.append("t%d: %6.1f us (d %.1e)" % (cfg[0], us, d))
        gf = 2.0 * H * W * cin * cout * 9 / 1e9
        print("%-28s %6.1f GF | %s" % (name, gf, "  ".join(out)), flush=True)
        # conv + statistics finalize as the frame chains them: separate launches (bn_partial_reduce + bn_finalize above 512 rows) for the
        # one-row-per-tile kernels, inside the launch for the persistent tiles (one row per workgroup, last workgroup finalizes)
        norm = nn.BatchNorm2d(cout).to("cuda:0")
        ss = torch.zeros(4 * cout, device="cuda:0")
        out = []
        for cfg in [c for c in TILES if c[0] in (94, 140, 141, 143)]:
            if cout > 64:
                break
            eng.tile_override[(cin, cout, 3, 1, 0)] = cfg
            def chain():
                raw, rows, shp = eng.conv(x, mod, L.PAD_REFLECT, 1, L.OUT_RAW_F32_NHWC, want_stats=True, fin=(norm, ss))
                if not eng.last_finalized:
                    st = eng.scratch("stats", rows * cout * 2)
                    groups = L.lib.v2v_bn_finalize_groups(rows)
                    ws = eng.scratch("bn_ws", groups * cout * 2, torch.float64) if groups > 0 else None
                    L.check(L.lib.v2v_bn_finalize(_ptr(st), rows, cout, H * W, _ptr(norm.weight.detach()), _ptr(norm.bias.detach()), norm.eps,
                                                  _ptr(ss), None, None, 0.1, _ptr(ws), _stream()), "bn_finalize")
                return rows
            try:
                us = timed(chain)
                out.append("t%d: %6.1f us (%d rows, finalize %s)" % (cfg[0], us, chain(), "in the launch" if eng.last_finalized else "separate"))
            except Exception as e:
                out.append("t%d: n/a (%s)" % (cfg[0], str(e)[:60]))
        if out:
            print("%-28s   + finalize | %s" % ("", "  ".join(out)), flush=True)
