#!/usr/bin/env python3
"""Per-layer roofline table from an op dump that carries its own census (bench.py --dump-ops, the `.hires.json` file of the
2048x1024 / 3-scale companion): every conv ALONE on the chip (eager single-stream replay, HIP events), priced against
max(FLOP / 2500 TFLOP/s, (bf16 input + bf16 weights + fp32 raw output, bf16 for fused pairs, fp32 planar for heads) / 6.29 TB/s).
    python scripts/per_layer_roofline_hires.py profiles/r03_c4_ops_hires_bf16.json > profiles/r03_c4_per_layer_roofline_hires.txt"""
import json
import sys
PEAK_TF, HBM_TBS = 2500.0, 6.29
ops = json.load(open(sys.argv[1]))
rows, other = [], {}
for o in ops:
    if "flops" not in o:
        other[o["op"]] = other.get(o["op"], 0.0) + o["ms"]
        continue
    m = o.get("members", 1)
    px_in, px_out = o["H"] * o["W"], o["OH"] * o["OW"]
    byts = m * (2.0 * (px_in * o["cin"] + o["cin"] * o["cout"] * o["KH"] ** 2) + (2.0 if m == 2 else 4.0) * px_out * o["cout"])
    t_mfma, t_hbm = o["flops"] / (PEAK_TF * 1e12) * 1e3, byts / (HBM_TBS * 1e12) * 1e3
    rows.append((o["ms"], o["label"], o, max(t_mfma, t_hbm), "mfma" if t_mfma >= t_hbm else "hbm"))
print("# per-layer roofline, every conv ALONE on the chip: %s" % sys.argv[1])
print("# bound = max(FLOP / %.0f TFLOP/s, algorithmic bytes / %.2f TB/s); gather-sum stems and non-conv ops are listed at the end" % (PEAK_TF, HBM_TBS))
print("%-34s %-26s %8s %8s %9s %9s %6s %5s  %s" % ("layer", "cin->cout k @HxW (in)", "GFLOP", "ms", "TFLOP/s", "bound_ms", "frac", "by", "tile,S,members"))
tb = tm = 0.0
for ms, label, o, bound, by in sorted(rows, key=lambda r: -r[0]):
    tb += bound; tm += ms
    print("%-34s %-26s %8.2f %8.4f %9.1f %9.4f %6.3f %5s  %s,%s,%s" % (label[:34], "%d->%d k%d @%dx%d" % (o["cin"], o["cout"], o["KH"], o["W"], o["H"]), o["flops"] / 1e9, ms,
                                                                       o["flops"] / ms / 1e9, bound, bound / ms, by, o.get("tile"), o.get("splitk"), o.get("members")))
print("# convolutions: sum alone %.3f ms, sum of bounds %.3f ms (%.1f %%)" % (tm, tb, 100.0 * tb / tm))
print("# other ops (ms, alone): " + ", ".join("%s %.3f" % kv for kv in sorted(other.items(), key=lambda kv: -kv[1])))
