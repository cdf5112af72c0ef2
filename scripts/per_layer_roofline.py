#!/usr/bin/env python3
"""Per-layer roofline table of the 512x256 label2city frame: joins the per-op HIP-event timings that bench.py dumped on the
GPU (--dump-ops: every kernel ALONE on the chip, eager single-stream replay) with the layer census of the same plan recorded
here on the CPU (record-only engine: shapes and algorithmic FLOP / bytes per conv), and prices each conv against
max(FLOP / MFMA peak, bytes / HBM).  No GPU needed.
    python scripts/per_layer_roofline.py profiles/r01_v20_ops_bf16.json > profiles/r01_v20_per_layer_roofline.txt"""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vid2vid_amd import networks as N
from vid2vid_amd.options import make_opt
from vid2vid_amd.models import create_model

PEAK_TF, HBM_TBS = 2500.0, 6.29          # bf16 dense MFMA peak; achievable HBM copy rate (MI355X_MICROARCH.md)

ops = json.load(open(sys.argv[1]))
N.set_record_only(True)
_so = sys.stdout
sys.stdout = sys.stderr
opt = make_opt(label_nc=35, use_instance=True, fg=True, use_real_img=True, random_init_ok=True, precision="bf16", gpu_ids=[])
m = create_model(opt)
H, W = 256, 512
m.inference(torch.randint(0, 35, (1, 3, 1, H, W)).float(), torch.zeros(1, 2, 3, H, W), torch.randint(0, 20, (1, 3, 1, H, W)).float())
sys.stdout = _so
log = {c["label"]: c for c in m._active_plan.conv_log}
rows, tot_ms, tot_bound = [], 0.0, 0.0
def census(c, fused):
    px_in = c["N"] * c["H"] * c["W"]
    px_out = c["N"] * c["OH"] * c["OW"]
    if c.get("onehot"):                      # gather-sum stem: uint8 code map in, bf16 NHWC out, the weight table once
        byts = 1.0 * px_in + 2.0 * c["cin"] * c["cout"] * c["KH"] * c["KW"] + 4.0 * px_out * c["cout"]
    else:                                    # bf16 input + weights; fp32 raw output, or bf16 (+ bf16 residual read) when the norm is fused
        byts = 2.0 * (px_in * c["cin"] + c["cin"] * c["cout"] * c["KH"] * c["KW"]) + (2.0 if fused else 4.0) * px_out * c["cout"] * c.get("convs", 1)
    return c["flops"] * c.get("convs", 1), byts


for o in ops:
    if o["op"] not in ("conv_igemm", "onehot_conv7x7"):
        continue
    parts = [log.get(l.strip()) for l in o["label"].split(" + ")] if " + " in o["label"] else [log.get(o["label"])]
    if any(c is None for c in parts):
        print("# no census for", o["label"], file=sys.stderr)
        continue
    fused = len(parts) == 2                  # the paired 1024->1024 launches carry the fused norm (bf16 output)
    flops = sum(census(c, fused)[0] for c in parts)
    byts = sum(census(c, fused)[1] for c in parts)
    c = dict(parts[0]); c["flops"] = flops; c["pair"] = len(parts)
    t_mfma = c["flops"] / (PEAK_TF * 1e12) * 1e3
    t_hbm = byts / (HBM_TBS * 1e12) * 1e3
    bound = max(t_mfma, t_hbm)
    rows.append((o["ms"], o["label"], c, byts, bound, "mfma" if t_mfma >= t_hbm else "hbm", o["tile"] or ("gather", "")))
    tot_ms += o["ms"]; tot_bound += bound
print("# per-layer roofline of the 512x256 label2city frame (bf16; one frame; every conv ALONE on the chip: %s)" % sys.argv[1])
print("# bound = max(FLOP / %.0f TFLOP/s, (bf16 input + bf16 weights + fp32 raw output [bf16 for the paired launches with the fused norm; uint8 codes in for the gather-sum stems, priced at their dense-conv FLOP] bytes) / %.2f TB/s)" % (PEAK_TF, HBM_TBS))
print("%-26s %-24s %8s %8s %9s %9s %6s %5s  %s" % ("layer", "cin->cout k/s @HxW", "GFLOP", "ms", "TFLOP/s", "bound_ms", "frac", "by", "tile,splitK"))
for ms, label, c, byts, bound, by, tile in sorted(rows, key=lambda r: -r[0]):
    shape = ("2x " if c.get("pair") == 2 else "") + "%d->%d k%d%s @%dx%d" % (c["cin"], c["cout"], c["KH"], ("T" if c["transposed"] else "") + "/s%d" % c["stride"], c["W"], c["H"])
    print("%-26s %-24s %8.2f %8.4f %9.1f %9.4f %6.3f %5s  %s" % (label[:26], shape, c["flops"] / 1e9, ms, c["flops"] / ms / 1e9, bound, bound / ms, by,
                                                             ",".join(str(t) for t in tile[:2])))
print("# sum over %d convs: %.3f ms measured alone, %.3f ms roofline bound -> %.1f %% of the per-layer roofline; in the graph (3 lanes) the whole frame takes less than the sum"
      % (len(rows), tot_ms, tot_bound, 100.0 * tot_bound / tot_ms))
