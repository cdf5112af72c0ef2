#!/usr/bin/env python3
"""Headline benchmark: synthesized frames/sec of Vid2VidModelG.inference() on MI355X.

Workload (BASELINE.json configs[1]): label2city 512x256, n_scales_spatial=1, --fg --use_instance,
ngf=128 / 9 blocks (411 M parameters, 2115 GFLOP per frame), batch 1 per sequence, one sequence
per GPU, bf16 storage + fp32 MFMA accumulate, random-init weights, seeded synthetic label /
instance / image sequences already resident in HBM.  A "step" is one generated frame: refresh
the plan's input buffers (device-to-device) + one replay of the frame plan (per-lane segment hipGraphs on three streams).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement), including
  roofline     -- the dominant kernel (the implicit-GEMM conv template instance that carries the
                  1024->1024 3x3 ResnetBlock convolutions): algorithmic FLOP per launch / average
                  launch duration, durations measured with HIP events around every launch of an
                  eager replay of the same plan on the same stream (v2v_plan_profile);
  cpu_baseline -- the CPU oracle (port of the reference's algorithm, torch CPU ops) timed on this
                  node's host cores on a bounded sample of the same workload (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3, "x3": 2500.0 / 3}   # dense MFMA peaks, MI355X_MICROARCH.md (x3: three bf16 products per product)
KERNEL_FAMILY = "conv_igemm"

LINE_BUDGET = 6000          # bytes of the ONE stdout line (the driver keeps a 10 KB stdout tail and parses its last line)
FULL_RECORD = "bench_full.json"


def _clip(s, n):
    return s if not isinstance(s, str) or len(s) <= n else s[:n - 3] + "..."


def compact_line(full):
    """The ONE stdout line: contract keys, the flat scalars, `config`, `roofline`, `cpu_baseline`, `timing` -- nothing nested deeper
    than that.  Every companion object (parity tables, hires / c1 / c4 / train legs, per-kernel tables) stays in the full record
    (`bench_full.json`).  Strings are clipped until the serialised line fits LINE_BUDGET."""
    contract = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    out = {k: full.get(k) for k in contract}
    for k, v in full.items():                                    # flat scalars (parity_value, hires_value, train_value, *_fp32_ok ...)
        if k not in out and (v is None or isinstance(v, (bool, int, float, str))):
            out[k] = v
    # per-leg scalars the nested objects carry (value + parity verdicts), flattened
    for leg in ("hires", "c1", "c4", "train", "train_hires", "train_c3", "fp32", "x3", "host_fed"):
        o = full.get(leg)
        if not isinstance(o, dict):
            continue
        if "error" in o:
            out[leg + "_error"] = _clip(o["error"], 120)
            continue
        if "value" in o:
            out.setdefault(leg + "_value", o["value"])
        p = o.get("parity")
        if isinstance(p, dict):
            for pk in ("fp32_ok", "x3_ok", "fp32_max_rel", "x3_max_rel", "bf16_max_rel"):
                if isinstance(p.get(pk), (bool, int, float)):
                    out.setdefault("%s_%s" % (leg, pk), p[pk])
            f32 = p.get("fp32")
            if isinstance(f32, dict):
                for pk in ("max_forward", "max_loss", "max_grad_norm", "max_grad_l2"):
                    if isinstance(f32.get(pk), (int, float)):
                        out.setdefault("%s_fp32_%s" % (leg, pk), f32[pk])
        if isinstance(o.get("cpu_baseline"), dict) and "value" in o["cpu_baseline"]:
            out.setdefault(leg + "_cpu_value", o["cpu_baseline"]["value"])
        if isinstance(o.get("flownet2"), dict) and "pairs_per_s" in o["flownet2"]:
            out.setdefault(leg + "_flownet2_pairs_per_s", o["flownet2"]["pairs_per_s"])
        if "frac_of_mfma_peak" in o:
            out.setdefault(leg + "_frac_of_mfma_peak", o["frac_of_mfma_peak"])
        if "peak_memory_gb" in o:
            out.setdefault(leg + "_peak_memory_gb", o["peak_memory_gb"])
        dm = o.get("dominant")
        if isinstance(dm, dict) and "frac" in dm:                 # the training leg's dominant kernel (roofline object of the train step)
            out.setdefault(leg + "_dominant_frac", dm["frac"])
            out.setdefault(leg + "_dominant_us", dm.get("avg_launch_us"))
            out.setdefault(leg + "_dominant_traffic", dm.get("traffic"))
    par = full.get("parity")
    if isinstance(par, dict):
        for pk in ("fp32_ok", "fp32_max_rel", "bf16_max_rel", "bf16_mean_rel", "tolerance_fp32"):
            if isinstance(par.get(pk), (bool, int, float)):
                out.setdefault("parity_" + pk, par[pk])
        if "error" in par:
            out["parity_error"] = _clip(par["error"], 120)
    t = full.get("timing")
    if isinstance(t, dict):
        out["timing"] = {k: t[k] for k in ("windows_ms_per_step", "statistic", "warmup_frames_run") if k in t}
    c = full.get("config")
    if isinstance(c, dict):
        out["config"] = {k: v for k, v in c.items() if v is None or isinstance(v, (bool, int, float, str))}
    r = full.get("roofline")
    if isinstance(r, dict):
        rr = {k: r.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic") if k in r}
        for k in ("avg_launch_us", "launches_per_frame", "flop_per_launch", "flop_per_step", "conv_launches_per_step", "measured", "resblock_1024_tflops"):
            if k in r:
                rr[k] = r[k]
        if isinstance(r.get("dominant"), dict):                   # --mode train: the step's dominant kernel beside the whole-step figure
            for k in ("kernel", "layer", "avg_launch_us", "achieved", "frac", "traffic", "algorithmic_bytes_per_launch"):
                if k in r["dominant"]:
                    rr["dominant_" + k] = r["dominant"][k]
        td = r.get("traffic_detail")
        if isinstance(td, dict):
            rr["algorithmic_bytes_per_launch"] = td.get("algorithmic_bytes_per_launch")
            rr["traffic_source"] = td.get("source")
        for k in ("eager", "in_graph_live", "in_graph_rocprof"):
            if isinstance(r.get(k), dict):
                rr[k + "_us"] = r[k].get("avg_launch_us")
                rr[k + "_frac"] = r[k].get("frac")
        if isinstance(r.get("frame_in_graph"), dict):
            rr["frame_frac"] = r["frame_in_graph"].get("frac")
        out["roofline"] = rr
    cb = full.get("cpu_baseline")
    out["cpu_baseline"] = None if not isinstance(cb, dict) else {k: cb[k] for k in ("value", "unit", "cores", "kind", "sample") if k in cb}
    if isinstance(full.get("leg_seconds"), dict):
        out["leg_seconds"] = full["leg_seconds"]
    out["full_record"] = FULL_RECORD
    # clip strings until the line fits
    for limit in (400, 240, 160, 100, 60):
        if len(json.dumps(out, allow_nan=False, default=_nonfinite)) <= LINE_BUDGET:
            break
        for d in (out, out.get("config") or {}, out.get("roofline") or {}, out.get("cpu_baseline") or {}):
            for k, v in list(d.items()):
                if isinstance(v, str):
                    d[k] = _clip(v, limit)
    return _finite(out)


def _nonfinite(o):
    return repr(o)


def _finite(o):
    """NaN / +-Infinity are not JSON: a strict parser must read the line."""
    if isinstance(o, float):
        return o if o == o and o not in (float("inf"), float("-inf")) else None
    if isinstance(o, dict):
        return {k: _finite(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_finite(v) for v in o]
    return o


def emit_record(full, stdout=None):
    """Full record -> bench_full.json (repo root and gpurun_out/ when it exists); compact line -> stdout (exactly one line, written last)."""
    stdout = stdout or sys.stdout
    full = _finite(full)
    text = json.dumps(full, allow_nan=False)
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        try:
            if os.path.isdir(d):
                with open(os.path.join(d, FULL_RECORD), "w") as f:
                    f.write(text + "\n")
        except OSError as ex:
            sys.stderr.write("bench: could not write %s in %s: %r\n" % (FULL_RECORD, d, ex))
    # (the full record is NOT echoed to stderr: a driver that merges the two streams into one 10 KB tail must still end in the line)
    sys.stderr.write("bench: full record (%d bytes) in %s\n" % (len(text), FULL_RECORD))
    sys.stderr.flush()
    comp = compact_line(full)
    line = json.dumps(comp, allow_nan=False)
    # never abort without a line: if numeric / flattened keys still push it over the budget, drop the optional parts --
    # leg timings, roofline extras, per-leg flats -- until it fits; the contract keys always go out
    contract = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline", "full_record")
    if len(line) > LINE_BUDGET + 2000:
        for victim in ("leg_seconds", "timing"):
            comp.pop(victim, None)
        if isinstance(comp.get("roofline"), dict):
            comp["roofline"] = {k: v for k, v in comp["roofline"].items() if k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_us")}
        line = json.dumps(comp, allow_nan=False)
        for k in [k for k in list(comp) if k not in contract]:
            if len(line) <= LINE_BUDGET + 2000:
                break
            comp.pop(k)
            line = json.dumps(comp, allow_nan=False)
    stdout.write(line.replace("\n", " ") + "\n")
    stdout.flush()


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32", "x3"])
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--height", type=int, default=256)
    ap.add_argument("--scales", type=int, default=1, help="n_scales_spatial (3 with --width 2048 --height 1024 = BASELINE configs[4] geometry, inference)")
    ap.add_argument("--dataset", default="label2city", choices=["label2city", "edge2face"],
                    help="edge2face with --width 512 --height 512 = BASELINE configs[3] (input_nc=15 raw maps, no fg tower)")
    ap.add_argument("--mode", default="infer", choices=["infer", "train"],
                    help="infer: Vid2VidModelG.inference() frames (the headline metric); train: train.py's inner loop "
                         "(G forward, FlowNet2, D / D_T, three backward passes + optimizer steps) in frames trained/s")
    ap.add_argument("--num-D", type=int, default=2, help="train: discriminator scales (3 with --width 1024 --scales 2 = configs[2])")
    ap.add_argument("--frames-total", type=int, default=6, help="train: n_frames_total of a sequence (scripts/street/train_512.sh)")
    ap.add_argument("--frames-per-gpu", type=int, default=2, help="train: max_frames_per_gpu = frames per chunk")
    ap.add_argument("--with-vgg", action="store_true", help="train: (default) include the VGG19 perceptual loss, random-init VGG19 weights")
    ap.add_argument("--no-vgg", action="store_true", help="train: the reference's --no_vgg recipe")
    ap.add_argument("--no-autotune", action="store_true", help="train: skip the per-shape tile search of the first chunks")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=2)
    ap.add_argument("--oracle-threads", type=int, default=16,
                    help="host threads of the CPU legs (cpu_baseline, the oracle sides of the parity legs); 0 = torch's default pool.  On the GPU box's "
                         "128-thread default the oracle is 7x SLOWER than on 16 (0.104 / 0.343 / 0.642 / 0.712 frames/s on 128 / 64 / 32 / 16 threads, "
                         "profiles/r06_v57_cpu_baseline_threads.txt): the baseline is quoted on the pool it runs fastest on")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--retune", action="store_true", help="ignore profiles/tune_cache.json and measure the tile selections in this run")
    ap.add_argument("--profile-frames", type=int, default=3)
    ap.add_argument("--no-train-line", action="store_true", help="infer: skip the short --mode train run reported under \"train\"")
    ap.add_argument("--dump-ops", default="", help="write the per-op timing table (json) here")
    ap.add_argument("--windows", type=int, default=5, help="timed windows of --steps frames each; `value` is the MEDIAN window")
    ap.add_argument("--min-warmup-s", type=float, default=0.5, help="warm up for at least this long (and at least --warmup frames) before the first window")
    ap.add_argument("--no-hires", action="store_true", help="infer: skip the 2048x1024 / 3-scale companion run reported under \"hires\"")
    ap.add_argument("--dry-run", action="store_true", help="plumbing check of the launch commands the driver uses (`--gpus N [--mode train]` under "
                    "torch.distributed.run): host CPU, gloo instead of RCCL, the library in dry-run mode (every launch argument-checked, nothing "
                    "executed).  Prints one line with data = \"dry-run\"; its value is a host rate, not a measurement")
    ap.add_argument("--ngf", type=int, default=128, help="generator width (128 = BASELINE's configuration; smaller only for --dry-run plumbing tests)")
    ap.add_argument("--train-hires-parity-compat", dest="no_train_hires_parity", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-train-hires-parity", dest="no_train_hires_parity", action="store_true",
                    help="(deprecated no-op: the 2048x1024 chunk's CPU-oracle parity is opt-in since round 5, --train-hires-parity)")
    ap.add_argument("--train-graph", action="store_true", help="train: replay each chunk kind as ONE captured hipGraph (vid2vid_amd/graphed.py) instead of "
                    "eager launches.  Measured equal (68.6 vs 69.0 ms per 512x256 chunk, profiles/r06_v3_traingraph.txt): the step is bound by the "
                    "device's kernel-to-kernel turnaround of ~4000 dependent launches, not by host time -- off by default")
    ap.add_argument("--no-train-graph", action="store_true", help="(default; kept for callers of the first round-6 builds)")
    ap.add_argument("--no-train-parity", action="store_true", help="train: skip the fp32 chunk-vs-oracle parity leg (outputs, losses, gradient norms)")
    ap.add_argument("--no-c4", action="store_true", help="infer: skip the BASELINE configs[3] leg (edge2face 512x512) reported under \"c4\"")
    ap.add_argument("--no-train-c3", action="store_true", help="infer: skip the BASELINE configs[2] geometry training leg (1024x512, 2 scales) reported under \"train_c3\"")
    ap.add_argument("--no-c1", action="store_true", help="infer: skip the literal BASELINE configs[0] leg (256x128 2-frame clip, GPU vs CPU oracle) reported under \"c1\"")
    ap.add_argument("--no-train-hires", action="store_true", help="infer: skip the 2048x1024 / 3-scale training chunk (configs[4] geometry on one GPU) reported under \"train_hires\"")
    ap.add_argument("--train-hires-parity", action="store_true", help="infer: also run the 2048x1024 training chunk's CPU-oracle parity inside the bench (minutes of host "
                    "time; the same check is tests/test_gpu_golden.py::test_full_width_training_chunk_2048x1024_s3_vs_oracle)")
    ap.add_argument("--n-gpus-gen", type=int, default=-1, help="train: the reference's --n_gpus_gen; smaller than --group-size selects the generator / "
                    "discriminator rank roles (vid2vid_amd/roles.py), e.g. --gpus 8 --group-size 8 --n-gpus-gen 6 = configs[4]'s 6 G + 2 D layout")
    ap.add_argument("--group-size", type=int, default=0, help="train: GPUs that share ONE sequence (len of the reference's --gpu_ids); 0 = 1 (plain data parallelism)")
    return ap.parse_args()


def train_parity(args, dev, local_rank, opt, modelG, modelD, flowNet):
    """Parity BEFORE timing ("losses and a gradient norm for train configs"): ONE chunk of train.py's inner loop
    at the benchmarked geometry in fp32 against the CPU oracle holding the same weights -- outputs, every loss of
    Vid2VidModelD.forward, and the gradients of G, D and D_T0 (norm and relative L2 distance of the whole flattened
    gradient); the same chunk through the bf16 kernels is reported beside it.  The chunk holds 3 frames so that temporal
    scale 0 is active (tD = 3); flow_ref / conf_ref come from this run's FlowNet2 and are inputs of both sides."""
    import torch
    from oracle import train_parity as TP
    from vid2vid_amd import synthetic
    from vid2vid_amd.options import make_opt
    from vid2vid_amd.models.vid2vid_model_G import Vid2VidModelG
    from vid2vid_amd.models.vid2vid_model_D import Vid2VidModelD
    H, W, S = args.height, args.width, args.scales
    nfl = 3 if S == 1 else 1                         # the CPU side costs ~20 s per frame at 512x256, ~55 s at 1024x512 / 2 scales
    tG = opt.n_frames_G
    nT = nfl + tG - 1
    lab, inst, frames = synthetic.label2city_sequence(nT, H, W, seed=4321, device=dev)
    A, I, B = lab.view(1, nT, 1, H, W), inst.view(1, nT, 1, H, W), frames
    with torch.no_grad():
        flow_ref, conf_ref = flowNet(B[:, tG - 1:], B[:, tG - 2:-1])
    flow_ref, conf_ref = flow_ref.detach().float(), conf_ref.detach().float()
    # FlowNet2 itself at THIS size against the oracle's FlowNet2 (its output enters the chunk's losses); the
    # network runs in the benchmarked precision here, so the 1e-3 gate applies to an fp32 run only (tests/test_gpu_golden.py
    # ::test_flownet2_at_baseline_size_vs_oracle is the fp32 check at 512x256 and 1024x512)
    fn2 = None
    try:
        from oracle import vid2vid_oracle as O
        fmod = flowNet.module if hasattr(flowNet, "module") else flowNet
        sdf = {k: v.detach().float().cpu() for k, v in fmod.flowNet.state_dict().items()}
        im1, im2 = B[0, tG - 1:].float().cpu(), B[0, tG - 2:-1].float().cpu()
        t0_ = time.perf_counter()
        with torch.no_grad():
            rf, rc = O.flow_and_conf(sdf, im1, im2)
        e = (flow_ref[0].cpu() - rf).abs() / (rf.abs() + rf.pow(2).mean().sqrt().item() + 1e-12)
        fn2 = {"flow_max_rel": float("%.3e" % e.max().item()), "flow_mean_rel": float("%.3e" % e.mean().item()),
               "conf_mismatch_fraction": float("%.3e" % (conf_ref[0].cpu() != rc).float().mean().item()),
               "precision": getattr(fmod, "precision", args.precision), "pairs": int(im1.shape[0]), "oracle_seconds": round(time.perf_counter() - t0_, 1),
               "note": "FlowNet (FlowNet2 + confidence) of the benchmarked run on the chunk's real frame pairs against oracle.flow_and_conf "
                       "holding the same weights; conf mismatches include pixels sitting on the 0.02 threshold"}
    except Exception as ex:
        fn2 = {"error": repr(ex)[:300]}
    srcG, srcD = modelG.module, modelD.module
    has_T = nfl >= opt.n_frames_D

    def pair(precision):
        o = make_opt(isTrain=True, label_nc=35, use_instance=True, fg=True, random_init_ok=True, loadSize=W, precision=precision,
                     gpu_ids=[local_rank], n_scales_spatial=S, num_D=args.num_D, n_frames_total=2 * max(nfl, 2),
                     max_frames_per_gpu=nfl, n_scales_temporal=1, no_vgg=args.no_vgg, niter_fix_global=0)
        G = Vid2VidModelG(); G.initialize(o)
        D = Vid2VidModelD(); D.initialize(o)
        for si in range(S):
            getattr(G, "netG%d" % si).load_state_dict(getattr(srcG, "netG%d" % si).state_dict())
        D.netD.load_state_dict(srcD.netD.state_dict())
        D.netD_T0.load_state_dict(srcD.netD_T0.state_dict())
        if not args.no_vgg:
            D.criterionVGG.vgg.load_state_dict(srcD.criterionVGG.vgg.state_dict())
        G.engine.refresh_weights()
        return o, G, D

    cpu = lambda m: {k: v.detach().float().cpu() for k, v in m.state_dict().items()}
    o32, G32, D32 = pair("fp32")
    ref = TP.oracle_chunk([cpu(getattr(G32, "netG%d" % si)) for si in range(S)], cpu(D32.netD), cpu(D32.netD_T0) if has_T else None,
                          A.cpu(), I.cpu(), B.cpu(), flow_ref.cpu(), conf_ref.cpu(),
                          n_down=o32.n_downsample_G, n_blocks=o32.n_blocks, n_blocks_local=o32.n_blocks_local, n_frames_load=nfl,
                          num_D=args.num_D, sd_vgg=None if args.no_vgg else cpu(D32.criterionVGG.vgg),
                          param_names=TP.param_names_of(G32, D32))
    # TEACHER-FORCED (as the inference parity above): every frame t > 0 starts from the oracle's own previous frames at every
    # scale, so all frames have the same inputs on both sides and the 1e-3 bar applies to every frame, every loss and the
    # gradient norms; the free-running chunk (each side feeds its own frames: ~1e-4 differences propagate) is reported beside it
    c32 = TP.compare(TP.hip_chunk(G32, D32, A, I, B, flow_ref, conf_ref, teacher=ref["fake_pyr"]), ref)
    free32 = TP.compare(TP.hip_chunk(G32, D32, A, I, B, flow_ref, conf_ref), ref) if nfl > 1 else None
    del G32, D32
    torch.cuda.empty_cache()
    c16 = None
    if args.precision != "fp32" and not getattr(args, "no_bf16_train_parity", False):
        _, G16, D16 = pair("bf16")
        c16 = TP.compare(TP.hip_chunk(G16, D16, A, I, B, flow_ref, conf_ref), ref)
        del G16, D16
        torch.cuda.empty_cache()
    # gradients: the norm's relative error is bounded by the relative L2 distance of the whole gradient (gated at 5e-3 here and in
    # tests/test_gpu_golden.py; the fp32 oracle itself sits 8.6e-4 from an fp64 oracle, profiles/r05_oracle_noise_floor_*.json); which
    # tiles the timing-based search picks moves the norm figure between 7.7e-5 and 1.2e-3 from run to run (r06_v50 / v62 / v68)
    ok = bool(c32["max_forward"] <= 1e-3 and c32["max_loss"] <= 1e-3 and c32["max_grad_norm"] <= 2.5e-3 and c32["max_grad_l2"] <= 5e-3
              and all(v["finite"] for v in c32["grads"].values()))
    return {"chunk": "label2city %dx%d, n_scales_spatial=%d, num_D=%d, %d frame%s (first chunk of a sequence), VGG %s, temporal scale 0 %s"
                     % (W, H, S, args.num_D, nfl, "s" if nfl > 1 else "", "off" if args.no_vgg else "on (this run's random-init VGG19)", "active" if has_T else "inactive"),
            "reference": "oracle/train_parity.py: CPU autograd of oracle/vid2vid_oracle.py, pinned to the reference's own chunk "
                         "(outputs, losses, complete gradients) by tests/golden/training_label2city_s2_32x64.npz",
            "measure": "forward: per pixel |got-ref| / (|ref| + rms(ref)); losses: |got-ref| / max(|ref|, 1e-3); gradients: "
                       "relative error of the norm and relative L2 distance of the whole flattened gradient per optimizer",
            "tolerance_fp32": {"forward_every_frame": 1e-3, "losses": 1e-3, "grad_norm": 2.5e-3, "grad_l2": 5e-3,
                               "note": "teacher-forced: every frame t > 0 of the product's chunk starts from the oracle's previous frames "
                                       "(same inputs on both sides); free_running_fp32 = the same chunk with each side feeding its own frames"},
            "fp32": c32, "fp32_ok": ok, "flownet2_vs_oracle": fn2,
            "free_running_fp32": None if free32 is None else {k: free32[k] for k in ("max_forward", "max_forward_frame0", "max_loss", "max_grad_norm", "max_grad_l2")},
            "bf16": c16, "oracle_seconds": round(ref["seconds"], 1)}


def train_dominant_kernel(dev, ch, h, w, reps=24):
    """conv_wgrad3x3_bf16_kernel (csrc/conv_wgrad.hip) on the ch -> ch 3x3 / reflect layer at h x w, accumulate = 1: average launch
    duration by HIP events on the launch stream (cold operands: a 384 MB memset in front of every launch), algorithmic FLOP per
    launch, fraction of the dense bf16 MFMA peak; `traffic` = HBM bytes per launch from the committed rocprofv3 --pmc passes of the
    same launch (scripts/gpu_r6.sh wgradpmc -> profiles/*_wgrad_traffic.json) when the shape matches."""
    import ctypes as C
    import glob
    import torch
    from vid2vid_amd import lib as L
    from vid2vid_amd.lib import lib, WgradDesc, check
    dy = torch.randn(1, h, w, ch, device=dev).bfloat16()
    x = torch.randn(1, h, w, ch, device=dev).bfloat16()
    zero = torch.zeros(256, dtype=torch.uint8, device=dev)
    grad = torch.zeros(ch, ch, 3, 3, device=dev)
    d = WgradDesc()
    d.p, d.q = dy.data_ptr(), x.data_ptr()
    d.N, d.OH, d.OW, d.QH, d.QW = 1, h, w, h, w
    d.rows, d.cols, d.p_stride, d.q_stride = ch, ch, ch, ch
    d.KH = d.KW = 3
    d.stride, d.pad, d.pad_mode = 1, 1, L.PAD_REFLECT
    d.dtype, d.accumulate = L.BF16, 1
    d.grad, d.zero_page = grad.data_ptr(), zero.data_ptr()
    nbytes = lib.v2v_conv_wgrad_workspace(C.byref(d))
    ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=dev)
    d.workspace = ws.data_ptr()
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    thrash = torch.empty(96 << 20, dtype=torch.float32, device=dev)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        thrash.zero_()
        a.record(); check(lib.v2v_conv_wgrad(C.byref(d), st), "wgrad"); b.record()
    torch.cuda.synchronize(dev)
    us = sorted(a.elapsed_time(b) for a, b in ev)[reps // 2] * 1e3
    flop = 2.0 * h * w * ch * ch * 9
    ach = flop / us / 1e6
    alg = (2 * h * w * ch * 2) + 2 * ch * ch * 9 * 4              # dY + X once (bf16), .grad read + written (fp32)
    out = {"kernel": "conv_wgrad3x3_bf16_kernel (nine taps per workgroup, gradient accumulated straight into .grad)",
           "layer": "%d -> %d 3x3 reflect at %dx%d" % (ch, ch, w, h), "bound": "mfma", "flop_per_launch": flop,
           "avg_launch_us": round(us, 2), "achieved": round(ach, 1), "peak": PEAK_TFLOPS["bf16"], "unit": "TFLOP/s",
           "frac": round(ach / PEAK_TFLOPS["bf16"], 4), "algorithmic_bytes_per_launch": alg, "traffic": None,
           "measured": "HIP events on the launch stream, alone on the chip, cold operands, median of %d" % reps}
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_wgrad_traffic.json")))
    if files:
        t = json.load(open(files[-1]))
        if t.get("layer") == [ch, ch, h, w] and t.get("hbm_bytes_per_launch"):
            out["traffic"] = t["hbm_bytes_per_launch"]
            out["traffic_source"] = os.path.relpath(files[-1], ROOT) + ": rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE, separate passes"
    return out


def run_train(args, dev, rank, world, local_rank, emit=True):
    """--mode train: the inner loop of the reference's train.py (:50-138) on one sequence per rank.

    One "step" = one chunk of n_frames_load frames: Vid2VidModelG.forward (autograd graph of v2v ops), FlowNet2 on
    the real frame pairs, Vid2VidModelD.forward for the image discriminator and every active temporal scale,
    get_losses, then zero_grad / backward / Adam step for G, D and each active D_T.  Sequences of n_frames_total
    frames are cycled (history reset at each sequence start, train.py:47-48).  Data-parallel over sequences for
    N > 1: per-rank replicas, gradients all-reduced over RCCL inside FusedAdam.step (parallel.GradSync).
    value = frames trained per second over all ranks."""
    import torch
    import torch.distributed as dist
    from vid2vid_amd import synthetic, parallel
    from vid2vid_amd.options import make_opt
    from vid2vid_amd.models import create_model
    from vid2vid_amd.models.models import create_optimizer

    H, W, S = args.height, args.width, args.scales
    # --group-size G --n-gpus-gen K (K < G): the reference's `--gpu_ids 0..G-1 --n_gpus_gen K` -- G ranks share ONE sequence, K of
    # them generate (k frames each per chunk), the rest run the discriminators / FlowNet2 (vid2vid_amd/roles.py); the
    # world holds world / G such sequence groups.  Default: every rank trains G and D on its own sequence (data parallel).
    group = max(int(args.group_size or 1), 1)
    if world % group != 0:
        raise SystemExit("bench.py: --group-size %d does not divide the world size %d" % (group, world))
    n_gen = args.n_gpus_gen if (group > 1 and args.n_gpus_gen > 0) else -1
    role_mode = group > 1 and 0 < n_gen < group
    dry = bool(getattr(args, "dry_run", False))
    opt = make_opt(isTrain=True, label_nc=35, use_instance=True, fg=True, random_init_ok=True, loadSize=W, ngf=getattr(args, "ngf", 128),
                   precision=args.precision, gpu_ids=list(range(group)) if role_mode else [0 if dry else local_rank],
                   n_gpus_gen=n_gen if role_mode else -1, n_scales_spatial=S, num_D=args.num_D,
                   n_frames_total=args.frames_total, max_frames_per_gpu=args.frames_per_gpu,
                   no_vgg=args.no_vgg, niter_fix_global=0)
    _stdout = sys.stdout
    sys.stdout = sys.stderr
    # seeded initialisation (G, D / D_T, FlowNet2, the stand-in VGG19): the parity leg below compares THIS draw with the oracle, and an
    # unseeded one made its gradient-norm figure vary with the run (reference norms 321 ... 917 over four runs, i.e. 6.8e-5 ... 1.1e-3
    # relative for the same ~1.0 of absolute error: profiles/r04_e2_default_line_bf16.json)
    torch.manual_seed(0)
    models = create_model(opt)            # role mode: roles.layout_from_opt -> the three role wrappers, per-role gradient groups
    modelG, modelD, flowNet, optimizer_G, optimizer_D, optimizer_D_T = create_optimizer(opt, models)
    layout = getattr(modelG, "layout", None)
    if role_mode and layout is None:
        raise SystemExit("bench.py: role mode requested but create_model returned plain wrappers")
    with torch.no_grad():
        for si in range(S):
            getattr(modelG.module, "netG%d" % si).model_final_flow[1].weight.mul_(0.1)
    gsync = None
    if world > 1 and not role_mode:
        gsync = parallel.sync_optimizers([optimizer_G, optimizer_D] + list(optimizer_D_T))
        gsync.timing = True
    seq_index = (rank // group) if role_mode else rank          # every rank of a sequence group loads the SAME sequence
    n_seqs = (world // group) if role_mode else world
    eng = modelG.module.engine
    tG, tD, t_scales = opt.n_frames_G, opt.n_frames_D, opt.n_scales_temporal
    n_frames_total, n_frames_load = opt.n_frames_total, modelG.module.n_frames_load
    n_seq_frames = n_frames_total + tG - 1
    lab, inst, frames = synthetic.label2city_sequence(n_seq_frames, H, W, seed=1234 + seq_index, device=dev)
    A_all = lab.view(1, n_seq_frames, 1, H, W)
    I_all = inst.view(1, n_seq_frames, 1, H, W)
    B_all = frames                                                  # (1, T, 3, H, W)

    def reshape(ts):
        return [None if t is None else t.contiguous().view(-1, t.size(2), t.size(3), t.size(4)) for t in ts]

    def loss_backward(loss, optimizer):                             # train.py:130-138
        optimizer.zero_grad()
        loss.backward()
        optimizer.step()

    state = {"i": 0, "fake_B_prev_last": None, "frames_all": (None, None, None, None), "loss": None}
    chunks_per_seq = max(n_frames_total // n_frames_load, 1)

    def step():
        i = state["i"]
        if i == 0:
            state["fake_B_prev_last"], state["frames_all"] = None, (None, None, None, None)
        te = i + n_frames_load + tG - 1
        input_A, input_B, inst_A = A_all[:, i:te], B_all[:, i:te], I_all[:, i:te]
        fake_B, fake_B_raw, flow, weight, real_A, real_Bp, fake_B_last = modelG(input_A, input_B, inst_A, state["fake_B_prev_last"])
        real_B_prev, real_B = real_Bp[:, :-1], real_Bp[:, 1:]
        flow_ref, conf_ref = flowNet(real_B, real_B_prev)
        fake_B_prev = modelG.module.compute_fake_B_prev(real_B_prev, state["fake_B_prev_last"], fake_B)
        state["fake_B_prev_last"] = fake_B_last
        losses = modelD(0, reshape([real_B, fake_B, fake_B_raw, real_A, real_B_prev, fake_B_prev, flow, weight, flow_ref, conf_ref]))
        loss_dict = dict(zip(modelD.module.loss_names, [torch.mean(x) for x in losses]))
        state["frames_all"], skipped = modelD.module.get_all_skipped_frames(
            state["frames_all"], real_B, fake_B, flow_ref, conf_ref, t_scales, tD, n_frames_load, i, flowNet)
        loss_dict_T = []
        for s in range(t_scales):
            if skipped[0][s] is not None:
                lt = modelD(s + 1, [f[s] for f in skipped])
                loss_dict_T.append(dict(zip(modelD.module.loss_names_T, [torch.mean(x) for x in lt])))
        loss_G, loss_D, loss_D_T, t_act = modelD.module.get_losses(loss_dict, loss_dict_T, t_scales)
        loss_backward(loss_G, optimizer_G)
        loss_backward(loss_D, optimizer_D)
        for s in range(t_act):
            loss_backward(loss_D_T[s], optimizer_D_T[s])
        state["loss"] = (loss_G.detach(), loss_D.detach(), t_act)
        state["i"] = i + n_frames_load if (i // n_frames_load + 1) < chunks_per_seq else 0

    # Round 6 (--train-graph): the chunk's whole launch sequence (G forward, FlowNet2, D / D_T, losses, the zero_grad / backward /
    # Adam triples) captured once per chunk kind (= position of the chunk in its sequence) and replayed as ONE hipGraph
    # (vid2vid_amd/graphed.py).  The eager step leaves the device idle 30 % of a chunk (profiles/r06_v1_train_by_grid.txt), but
    # the replay takes the same time: the idle share is the device's turnaround between dependent launches, not the host.
    # Single process only: RCCL collectives and the role mode's point-to-point transfers stay on the eager path.
    from vid2vid_amd.graphed import ChunkGraphs
    use_graph = bool(world == 1 and not role_mode and getattr(args, "train_graph", False) and not args.no_train_graph)
    graphs = ChunkGraphs([optimizer_G, optimizer_D] + list(optimizer_D_T), device=dev, enabled=False)
    step_eager = step
    graph_logs = {}                                                 # chunk kind -> (conv_log entries, FlowNet2 flops, FlowNet2 convs) of one replay
    fnm = flowNet.module if hasattr(flowNet, "module") else flowNet

    def step():
        if not graphs.enabled:
            return step_eager()
        key = state["i"]
        if key not in graphs.graphs:
            n0, f0, c0 = len(eng.conv_log), fnm.flops_launched, fnm.convs_launched
            graphs.step(key, step_eager, lambda: dict(state), state.update)
            graph_logs[key] = (list(eng.conv_log[n0:]), fnm.flops_launched - f0, fnm.convs_launched - c0)
        else:
            graphs.step(key, step_eager, lambda: dict(state), state.update)
            log, fl, nc = graph_logs[key]
            eng.conv_log.extend(log)                               # what this replay launched (the roofline's FLOP count)
            fnm.flops_launched += fl
            fnm.convs_launched += nc

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    parity = None
    if rank == 0 and world == 1 and not args.no_train_parity:
        try:
            parity = train_parity(args, dev, local_rank, opt, modelG, modelD, flowNet)
        except Exception as ex:                                     # reported, never silently dropped
            import traceback
            traceback.print_exc()
            parity = {"error": repr(ex)[:400]}

    # Inside the default command this leg runs behind a dozen other models (inference, 2048x1024, the configs[0] / [3] clips, their
    # fp32 / x3 twins and the CPU oracle's tensors): millions of live Python objects.  A training chunk allocates ~10^5 objects (autograd
    # nodes, views, descriptors), so the cyclic collector's full passes -- whose cost grows with everything alive -- land inside the
    # timed chunks and the host falls behind the device (27.4 frames trained/s in the default line against 39-40 for the same leg
    # alone, profiles/r06_v22_bench_default_line.json vs r06_v23).  What is alive now is not garbage of this leg: park it.
    import gc
    gc.collect()
    gc.freeze()
    # tile search on the first sequence (every conv shape of the step: forward, backward-data), then plain warm-up
    eng.autotune = not args.no_autotune
    t_tune = time.perf_counter()
    for _ in range(chunks_per_seq):
        step()
    eng.autotune = False
    barrier()
    t_tune = time.perf_counter() - t_tune
    t_capture = 0.0
    if use_graph:
        for _ in range(chunks_per_seq):                             # one more eager sequence: steady state (every packed weight stale as in
            step()                                                  # any later chunk), state["i"] back at the start of a sequence
        barrier()
        t_capture = time.perf_counter()
        graphs.enabled = True
        for o in graphs.optimizers:
            o.make_capturable()
        for _ in range(chunks_per_seq):                             # capture every chunk kind, in sequence order
            step()
        barrier()
        t_capture = time.perf_counter() - t_capture
    for _ in range(args.warmup):
        step()
    eng.conv_log = []
    fn_flops0, fn_convs0 = flowNet.module.flops_launched, flowNet.module.convs_launched
    barrier()
    torch.cuda.reset_peak_memory_stats(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    t_host = time.perf_counter() - t0                           # the host has issued every launch of the timed chunks (no wait in between)
    barrier()
    elapsed = time.perf_counter() - t0
    peak_alloc = torch.cuda.max_memory_allocated(dev)
    free_b, total_b = torch.cuda.mem_get_info(dev)
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = tt.item()
    fps = n_seqs * args.steps * n_frames_load / elapsed          # frames of all sequences (main() refuses world != --gpus)
    loss_G, loss_D, t_act = state["loss"]
    finite = bool(torch.isfinite(loss_G).all().item() and torch.isfinite(loss_D).all().item())
    # FlowNet2 alone (frozen, inference: models/flownet2_pytorch via models/flownet.py:26-44): frame pairs per second
    flownet_line = None
    if rank == 0 and not role_mode:
        rb, rbp = B_all[:, 1:1 + n_frames_load], B_all[:, :n_frames_load]
        f0, c0 = flowNet.module.flops_launched, flowNet.module.convs_launched
        for _ in range(2):
            flowNet(rb, rbp)
        torch.cuda.synchronize(dev)
        f0, c0 = flowNet.module.flops_launched, flowNet.module.convs_launched
        tf0 = time.perf_counter()
        nrep = 10
        for _ in range(nrep):
            flowNet(rb, rbp)
        torch.cuda.synchronize(dev)
        tf = time.perf_counter() - tf0
        fl = flowNet.module.flops_launched - f0
        flownet_line = {"pairs_per_s": round(nrep * n_frames_load / tf, 2), "ms_per_pair": round(tf / (nrep * n_frames_load) * 1e3, 3),
                        "gflop_per_pair": round(fl / (nrep * n_frames_load) / 1e9, 1),
                        "tflops": round(fl / tf / 1e12, 2), "frac_of_mfma_peak": round(fl / tf / 1e12 / PEAK_TFLOPS[args.precision], 4),
                        "convs_per_pair": (flowNet.module.convs_launched - c0) // (nrep * n_frames_load),
                        "note": "FlowNet2 (C + S + S + SD + fusion) on %dx%d pairs, hipGraph replay, correlation on the matrix pipe (v2v_correlation_nhwc)" % (W, H)}

    # ---- the training step's dominant kernel: the weight gradient of the 8*ngf -> 8*ngf 3x3 ResnetBlock layers
    # (72 launches per 512x256 chunk, as many FLOP as the forward and backward-data launches of those layers), measured here with
    # HIP events on the stream it is launched on, alone on the chip, accumulating into .grad as in the step ----
    dominant = None
    if rank == 0 and not dry and args.precision == "bf16" and not role_mode:
        try:
            dominant = train_dominant_kernel(dev, 8 * opt.ngf, H >> opt.n_downsample_G, W >> opt.n_downsample_G)
        except Exception as ex:
            dominant = {"error": repr(ex)[:300]}
    # what travelled between the ranks, and how much of it the in-backward buckets hid (parallel.GradSync.overlap_report: HIP event
    # timestamps of every bucket's all-reduce against the end of the backward pass that produced it)
    sync_report = gsync.overlap_report() if gsync is not None else None
    if role_mode:
        collective_note = ("role mode: RCCL point-to-point of frames / gradients between generator and discriminator ranks, all-reduce of "
                           "the flat G gradient over the generator ranks (vid2vid_amd/roles.py)")
    elif world > 1:
        collective_note = ("bucketed all-reduce (sum, 1/world inside the Adam kernel) of the flat fp32 gradient buffers of G, D and each D_T "
                           "per chunk, G's buckets sent from inside its backward pass (vid2vid_amd/parallel.py); %d-byte buckets"
                           % (gsync.bucket_elems * 4))
    else:
        collective_note = "none (one process)"
    gc.unfreeze()
    sys.stdout = _stdout
    if rank == 0:
        log = list(eng.conv_log)
        by_kind = {}
        for c in log:
            k = c.get("kind", "fwd")
            by_kind[k] = by_kind.get(k, 0.0) + c["flops"]
        by_kind["flownet2_fwd"] = flowNet.module.flops_launched - fn_flops0      # hipGraph replays: not in the eager log
        n_launch = len(log) + (flowNet.module.convs_launched - fn_convs0)
        flop_step = sum(by_kind.values()) / args.steps
        peak = PEAK_TFLOPS[args.precision]
        ach = flop_step * args.steps / elapsed / 1e12
        roofline = {
            "bound": "mfma", "kernel": "whole training step: every conv forward / backward-data / backward-weight launch "
                                       "(G, FlowNet2, D, D_T); see profiles/ for the rocprofv3 per-kernel split",
            "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": None,
            "flop_per_step": flop_step, "conv_launches_per_step": n_launch // args.steps, "dominant": dominant,
            "gflop_per_step_by_kind": {k: round(v / args.steps / 1e9, 1) for k, v in sorted(by_kind.items())},
            "note": ("algorithmic conv FLOP of one chunk / wall time of one chunk; the chunk's whole launch sequence is replayed as one "
                     "hipGraph per chunk kind (vid2vid_amd/graphed.py)" if graphs.enabled else
                     "algorithmic conv FLOP of one chunk / wall time of one chunk (host launch time included: the "
                     "training step is an eager autograd graph of v2v custom ops, not a hipGraph)")
                    + ("; role mode: only THIS rank's launches (generator rank 0) are counted" if role_mode else ""),
        }
        out = {
            "metric": "frames trained/sec (train.py inner loop, %dx%d)" % (W, H),
            "value": round(fps, 3), "unit": "frames/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.precision, "data": "dry-run (no device: host rate of the launch sequence)" if dry else "synthetic",
            "config": {"workload": "label2city %dx%d train, n_scales_spatial=%d num_D=%d n_scales_temporal=%d, --fg --use_instance%s, "
                                   "n_frames_total=%d, %d frames per chunk, niter_fix_global=0 (all scales train), G %.1fM + D %.1fM "
                                   "params random-init, FlowNet2 random-init, 1 sequence per GPU"
                                   % (W, H, S, opt.num_D, t_scales, " --no_vgg" if args.no_vgg else "", n_frames_total,
                                      n_frames_load, sum(q.numel() for q in modelG.module.parameters()) / 1e6,
                                      sum(q.numel() for q in modelD.module.parameters()) / 1e6),
                       "frames_per_step": n_frames_load, "active_temporal_scales_last_step": int(t_act),
                       "autotune_s": round(t_tune, 1), "host_issue_ms_per_step": round(t_host / args.steps * 1e3, 2),
                       "train_graph": {"enabled": bool(graphs.enabled), "chunk_kinds": len(graphs.graphs), "capture_s": round(t_capture, 1),
                                       "replays": graphs.replays},
                       "parallelism": ("%d sequence group(s) x (%d generator + %d discriminator ranks): frames of a chunk split over the generator "
                                       "ranks, RCCL point-to-point for frames / gradients, all-reduce per role (vid2vid_amd/roles.py)"
                                       % (n_seqs, n_gen, group - n_gen)) if role_mode else
                                      "dp%d over sequences (RCCL all-reduce of flat gradients per optimizer)" % args.gpus,
                       "world_size": world, "sequences": n_seqs,
                       "backend": (dist.get_backend() if (world > 1 and dist.is_initialized()) else "none"),
                       "collective": collective_note, "grad_sync": sync_report, "dry_run": dry,
                       "grad_sync_buckets_inside_backward": None if sync_report is None else sync_report["buckets_sent_inside_backward"],
                       "grad_sync_buckets_at_step": None if sync_report is None else sync_report["buckets_sent_at_step"],
                       "grad_sync_allreduce_ms": None if sync_report is None else sync_report["allreduce_ms"],
                       "grad_sync_hidden_ms": None if sync_report is None else sync_report["hidden_ms"],
                       "loss_G": round(float(loss_G), 4), "loss_D": round(float(loss_D), 4), "output_finite": finite,
                       "peak_memory_gb": round(peak_alloc / 2 ** 30, 2),
                       "device_memory_in_use_gb": round((total_b - free_b) / 2 ** 30, 2),
                       "peak_memory_note": "torch allocator peak over the timed chunks (parameters, flat gradient / moment buffers, "
                                           "activations saved for backward, packed weights, scratch); device_memory_in_use = hipMemGetInfo "
                                           "after the run (allocator cache and the library's pools included)"},
            "parity": parity, "roofline": roofline, "flownet2": flownet_line, "cpu_baseline": None,
        }
        if emit:
            emit_record(out, _stdout)
        return out
    return None


def main():
    t_main = time.perf_counter()
    args = parse()
    import torch
    import torch.distributed as dist
    if int(os.environ.get("V2V_ORACLE_THREADS", "0")) > 0:
        args.oracle_threads = int(os.environ["V2V_ORACLE_THREADS"])
    if args.oracle_threads > 0 and torch.get_num_threads() > args.oracle_threads:
        torch.set_num_threads(args.oracle_threads)      # (nothing of the product runs on torch's CPU pool)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU, the command line the
        # driver uses) instead of timing one GPU and multiplying by N
        import socket
        import subprocess
        sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but the launched world size is %d (WORLD_SIZE); refusing to report a rate for GPUs "
                         "that did not run\n" % (args.gpus, world))
        sys.exit(2)
    dry = bool(args.dry_run)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo" if dry else "nccl", rank=rank, world_size=world)
    if dry:
        # no device: the launch sequence, the process group, the sharding of sequences over ranks and the JSON line are what is checked
        from vid2vid_amd import networks as N_
        N_.set_record_only(True)
        dev = torch.device("cpu")
        for name, fn in (("synchronize", lambda *a, **k: None), ("empty_cache", lambda *a, **k: None),
                         ("reset_peak_memory_stats", lambda *a, **k: None), ("max_memory_allocated", lambda *a, **k: 0),
                         ("mem_get_info", lambda *a, **k: (0, 0))):
            setattr(torch.cuda, name, fn)
        for k_ in ("no_cpu_baseline", "no_train_line", "no_train_hires", "no_c1", "no_c4", "no_train_c3", "no_hires", "no_train_parity", "no_autotune"):
            setattr(args, k_, True)
    else:
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)

    from vid2vid_amd import synthetic
    from vid2vid_amd.options import make_opt
    from vid2vid_amd.models import create_model

    H, W = args.height, args.width
    torch.manual_seed(0)
    if args.mode == "train":
        run_train(args, dev, rank, world, local_rank)
        if world > 1:
            dist.destroy_process_group()
        return
    # ---- tuning cache: replay the committed selections (the very tiles profiles/ describes) unless --retune ----
    tune_src = None
    if not os.environ.get("V2V_TUNE_CACHE") and not args.retune:
        committed = os.path.join(ROOT, "profiles", "tune_cache.json")
        if os.path.exists(committed):
            import shutil, tempfile
            tmp = os.path.join(tempfile.gettempdir(), "v2v_tune_replay_%d_%d.json" % (os.getpid(), rank))
            shutil.copyfile(committed, tmp)              # the engine appends new shapes to its cache: never touch the tracked file
            os.environ["V2V_TUNE_CACHE"] = tmp
            tune_src = "profiles/tune_cache.json"
    # N > 1: rank 0 runs the tile searches, the other ranks replay its selections (vid2vid_amd/parallel.py)
    from vid2vid_amd import parallel
    release_tuning = parallel.shared_tuning_cache(rank, world)
    face = args.dataset == "edge2face"

    def build_model(precision):
        if face:      # scripts/face/test_512.sh geometry: 15 raw input maps per frame, no instance map, no fg tower
            o = make_opt(label_nc=0, input_nc=15, use_instance=False, fg=False, use_real_img=True, random_init_ok=True,
                         dataroot="datasets/face/", loadSize=W, precision=precision, gpu_ids=[] if dry else [local_rank],
                         n_scales_spatial=args.scales, ngf=args.ngf)
        else:
            o = make_opt(label_nc=35, use_instance=True, fg=True, use_real_img=True, random_init_ok=True,
                         loadSize=W, precision=precision, gpu_ids=[] if dry else [local_rank], n_scales_spatial=args.scales, ngf=args.ngf)
        o.use_graph = not args.no_graph
        return o, create_model(o)

    sys.stdout.flush()
    _stdout = sys.stdout
    sys.stdout = sys.stderr                      # keep stdout for the single JSON line
    opt, model = build_model(args.precision)
    with torch.no_grad():
        for si in range(args.scales):                        # flows of a few px
            getattr(model, "netG%d" % si).model_final_flow[1].weight.mul_(0.1)
    tG = opt.n_frames_G
    L = 16                                       # resident sequence length, cycled
    if face:
        A, frames = synthetic.edge2face_sequence(L + tG, H, W, seed=1234 + rank, device=dev)
        lab = inst = I = None
    else:
        lab, inst, frames = synthetic.label2city_sequence(L + tG, H, W, seed=1234 + rank, device=dev)
        A = lab.view(1, L + tG, 1, H, W)
        I = inst.view(1, L + tG, 1, H, W)

    def run_step(m, t):
        k = t % L
        return m.inference(A[:, k:k + tG], frames[:, :tG - 1] if t == 0 else None, None if face else I[:, k:k + tG])

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    def timed_fps(m, steps, warmup, windows=1, min_warm_s=0.0):
        """Warm up for >= `warmup` frames AND >= min_warm_s seconds (clocks ramp after the idle CPU legs), then time
        `windows` windows of EXACTLY `steps` frames, each bracketed by barrier + synchronize on both sides and reduced
        with MAX over the ranks.  Returns (median window, all windows, warm-up frames)."""
        m.fake_B_prev = None
        t, tw = 0, time.perf_counter()
        while t < warmup or (time.perf_counter() - tw) < min_warm_s:
            run_step(m, t)
            t += 1
            if t % 8 == 0:
                torch.cuda.synchronize(dev)          # the host must not run ahead: the warm-up time is GPU time
        els = []
        for _ in range(max(windows, 1)):
            barrier()
            t0 = time.perf_counter()
            for _k in range(steps):
                run_step(m, t)
                t += 1
            barrier()
            el = time.perf_counter() - t0
            if world > 1:
                tt = torch.tensor([el], device=dev, dtype=torch.float64)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                el = tt.item()
            els.append(el)
        return sorted(els)[len(els) // 2], els, t - steps * len(els)

    model.fake_B_prev = None
    run_step(model, 0)                           # builds the frame plan (tile searches, graph): never inside the timed region
    torch.cuda.synchronize(dev)
    release_tuning()                             # N > 1, rank 0: the other ranks may now build with its selections
    if dry:
        el, els, nwarm = timed_fps(model, args.steps, args.warmup)
        sys.stdout = _stdout
        if rank == 0:
            emit_record({
                "metric": "synthesized frames/sec (Vid2VidModelG.inference, %dx%d)" % (W, H), "value": round(world * args.steps / el, 3),
                "unit": "frames/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(el / args.steps * 1e3, 4),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "dry-run (no device: host rate of the launch sequence)",
                "config": {"workload": "label2city %dx%d, n_scales_spatial=%d, ngf=%d, one sequence per rank" % (W, H, args.scales, args.ngf),
                           "world_size": world, "sequences": world, "backend": dist.get_backend() if world > 1 else "none",
                           "parallelism": "replicas only: one process per GPU, one sequence per rank, no data-path collective "
                                          "(the timing all-reduce(MAX) and the barrier are the only collectives)",
                           "collective": "none in the data path (inference replicas); barrier + all_reduce(MAX) of the window time",
                           "dry_run": True, "launches_per_frame": model._active_plan.plan.num_ops},
                "roofline": None, "cpu_baseline": None}, _stdout)
        if world > 1:
            dist.destroy_process_group()
        return

    # ---------------- host-fed rate: uint8 labels + int32 instance ids from pinned host memory every frame ----
    # (a side figure, measured before the CPU oracle runs: after 128 oracle threads have been busy the launching thread of this
    # loop is measurably slower -- 258 instead of 300 frames/s -- although the GPU-resident `value` below is not affected)
    host_fed = None
    if rank == 0 and world == 1 and not face:
        try:
            lab8, inst32 = lab.to(torch.uint8).cpu().pin_memory(), inst.to(torch.int32).cpu().pin_memory()
            # double-buffered staging: frame t+1's maps are uploaded on a copy stream while frame t's graph runs
            dA = [torch.empty(1, tG, 1, H, W, dtype=torch.uint8, device=dev) for _ in range(2)]
            dI = [torch.empty(1, tG, 1, H, W, dtype=torch.int32, device=dev) for _ in range(2)]
            copy_stream = torch.cuda.Stream(device=dev)
            ready = [torch.cuda.Event(), torch.cuda.Event()]
            consumed = [torch.cuda.Event(), torch.cuda.Event()]

            def upload(t):
                b, k = t & 1, t % L
                with torch.cuda.stream(copy_stream):
                    copy_stream.wait_event(consumed[b])          # the frame that last read this buffer has taken its copy
                    dA[b].view(tG, H, W).copy_(lab8[k:k + tG], non_blocking=True)
                    dI[b].view(tG, H, W).copy_(inst32[k:k + tG], non_blocking=True)
                    ready[b].record(copy_stream)

            def host_step(t):
                b = t & 1
                torch.cuda.current_stream(dev).wait_event(ready[b])
                out_ = model.inference(dA[b], frames[:, :tG - 1] if t == 0 else None, dI[b])
                consumed[b].record(torch.cuda.current_stream(dev))
                upload(t + 1)
                return out_

            for b_ in range(2):
                consumed[b_].record(torch.cuda.current_stream(dev))
            upload(0)
            model.fake_B_prev = None
            for t in range(3):
                host_step(t)                       # builds the uint8 frame plan
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for t in range(3, 3 + args.steps):
                host_step(t)
            torch.cuda.synchronize(dev)
            el_h = time.perf_counter() - t0
            host_fed = {"value": round(args.steps / el_h, 3), "unit": "frames/s", "ms_per_step": round(el_h / args.steps * 1e3, 4),
                        "h2d_bytes_per_frame": tG * H * W * 5,
                        "note": "every frame uploads its %d label maps (uint8) and instance maps (int32) from pinned host memory, double-buffered "
                                "on a copy stream (the reference's loader hands over host tensors, test.py:37-45); PCIe-inclusive, never `value`" % tG}
            model.fake_B_prev = None
            run_step(model, 0)                      # back on the resident fp32-encoded plan for the profile below
            torch.cuda.synchronize(dev)
        except Exception as ex:
            host_fed = {"error": repr(ex)[:300]}

    # ---------------- parity BEFORE timing: oracle vs the fp32 path and vs the benchmarked bf16 path ----------
    # rank 0 at N=1 only: the CPU oracle (also the cpu_baseline) generates PAR_FRAMES frames of the same seeded workload;
    # an fp32 model holding the same weights and the benchmarked model replay them.  Every output of the finest scale is
    # compared per pixel: |got - ref| / (|ref| + rms(ref)), maximum and mean over the tensor (tests/util.py).
    cpu = None
    parity = None
    fp32_line = None
    x3_line = None
    do_cpu = rank == 0 and args.gpus == 1 and not args.no_cpu_baseline
    if do_cpu:
        from oracle import vid2vid_oracle as O
        ncores = torch.get_num_threads()     # torch's default pool (one thread per physical core); SMT threads measured 9x slower
        S = args.scales
        sds = [{k: v.detach().float().cpu() for k, v in getattr(model, "netG%d" % s).state_dict().items()} for s in range(S)]
        fc = frames.cpu()
        nfr = 1 + args.cpu_frames
        if face:
            orc = O.InferenceOracle(sds, 0, False, False, [], opt.n_downsample_G, opt.n_blocks, opt.n_blocks_local)
            Ac = A.cpu()
            cpu_in = lambda t: (Ac[:, t:t + tG], fc[:, :tG - 1] if t == 0 else None, None)
        else:
            orc = O.InferenceOracle(sds, 35, True, True, [26], opt.n_downsample_G, opt.n_blocks, opt.n_blocks_local)
            lc, ic = lab.cpu(), inst.cpu()
            cpu_in = lambda t: (lc[t:t + tG].view(1, tG, 1, H, W), fc[:, :tG - 1] if t == 0 else None, ic[t:t + tG].view(1, tG, 1, H, W))
        refs, prev_states, cpu_s = [], [], 0.0
        for t in range(nfr):
            prev_states.append(None if orc.fake_B_prev is None else [q.clone() for q in orc.fake_B_prev])
            c0 = time.perf_counter()
            fake_ref, _ = orc.step(*cpu_in(t))
            if t > 0:
                cpu_s += time.perf_counter() - c0            # frame 0 is the warm-up
            refs.append(dict(fake_B=fake_ref.clone(), raw=orc.last["raw0"].clone(), flow=orc.last["flow0"].clone(),
                             weight=orc.last["weight0"].clone()))
        model_name = ""
        try:
            for line in open("/proc/cpuinfo"):
                if line.startswith("model name"):
                    model_name = line.split(":", 1)[1].strip(); break
        except OSError:
            pass
        cpu = {"value": round(args.cpu_frames / cpu_s, 4), "unit": "frames/s", "cores": ncores, "kind": "port",
               "sample": "%d frames (after 1 warm-up) of the same %dx%d workload, fp32, oracle/vid2vid_oracle.py on %d threads of %s (--oracle-threads: "
                         "the pool size it runs fastest on; the default 128-thread pool is 7x slower)"
                         % (args.cpu_frames, W, H, ncores, model_name or "host CPU")}

        def errors_of(m):
            """Every frame starts from the REFERENCE's previous frames (same inputs on both sides, as north_star words
            the bar): frame 0 from the given real frames, frame t > 0 from the oracle's own fake_B_prev pyramid.  The
            free-running drift of the last frame (each side fed by its own outputs) is reported separately."""
            m.fake_B_prev = None
            worst = {}
            for t in range(nfr):
                if t > 0:
                    for si in range(S):
                        m._active_plan.prev[si].copy_(prev_states[t][si])
                fake, _ = run_step(m, t)
                fpm = m._active_plan
                got = dict(fake_B=fake, raw=fpm.out["raw0"], flow=fpm.out["flow0"], weight=fpm.out["weight0"])
                for k, ref in refs[t].items():
                    g = got[k].detach().float().cpu()
                    rms = ref.pow(2).mean().sqrt().item() + 1e-12
                    e = (g - ref).abs() / (ref.abs() + rms)
                    w = worst.setdefault(k, {"max_rel": 0.0, "mean_rel": 0.0, "finite": True})
                    w["max_rel"] = max(w["max_rel"], e.max().item())
                    w["mean_rel"] = max(w["mean_rel"], e.mean().item())
                    w["finite"] = w["finite"] and bool(torch.isfinite(g).all().item())
            return {k: {"max_rel": float("%.3e" % v["max_rel"]), "mean_rel": float("%.3e" % v["mean_rel"]), "finite": v["finite"]}
                    for k, v in worst.items()}

        def drift_of(m):                           # free-running: the model's own frames feed the next one
            m.fake_B_prev = None
            for t in range(nfr):
                fake, _ = run_step(m, t)
            ref = refs[nfr - 1]["fake_B"]
            g = fake.detach().float().cpu()
            e = (g - ref).abs() / (ref.abs() + ref.pow(2).mean().sqrt().item() + 1e-12)
            return {"max_rel": float("%.3e" % e.max().item()), "mean_rel": float("%.3e" % e.mean().item())}

        opt32, model32 = (opt, model) if args.precision == "fp32" else build_model("fp32")
        if model32 is not model:
            for s in range(S):
                getattr(model32, "netG%d" % s).load_state_dict(getattr(model, "netG%d" % s).state_dict())
            model32.engine.refresh_weights()
        e32 = errors_of(model32)
        e16 = errors_of(model) if args.precision != "fp32" else None          # the benchmarked model (bf16, or x3 with --precision x3)
        drift = {"fp32_fake_B_frame%d" % (nfr - 1): drift_of(model32)}
        if args.precision == "bf16":
            drift["bf16_fake_B_frame%d" % (nfr - 1)] = drift_of(model)
        parity = {"frames": nfr, "reference": "oracle/vid2vid_oracle.py (CPU fp32 restatement pinned to the reference's own outputs, tests/golden)",
                  "measure": "per pixel |got-ref| / (|ref| + rms(ref)); max and mean over each tensor, worst frame; every frame "
                             "starts from the reference's own previous frames (same inputs)",
                  "tolerance_fp32": 1e-3,
                  "fp32": e32, "fp32_max_rel": max(v["max_rel"] for v in e32.values()),
                  "fp32_ok": bool(max(v["max_rel"] for v in e32.values()) <= 1e-3 and all(v["finite"] for v in e32.values())),
                  "benchmarked_precision": args.precision,
                  "bf16": e16 if args.precision == "bf16" else None,
                  "bf16_max_rel": None if (e16 is None or args.precision != "bf16") else max(v["max_rel"] for v in e16.values()),
                  "bf16_mean_rel": None if (e16 is None or args.precision != "bf16") else max(v["mean_rel"] for v in e16.values()),
                  "x3": e16 if args.precision == "x3" else None,
                  "free_running_drift": drift,
                  "note": "bf16 storage is a throughput mode: its error is reported, not gated at 1e-3 (the reference's own "
                          "bf16-autocast differs from its fp32 by 2e-2, BASELINE.md section 2)"}
        if model32 is not model:                   # companion figure: the path that carries the 1e-3 parity, same workload
            n32 = max(args.steps // 2, 5)
            el32, _, _ = timed_fps(model32, n32, 2, windows=3, min_warm_s=0.3)
            fp32_line = {"value": round(n32 / el32, 3), "unit": "frames/s", "ms_per_step": round(el32 / n32 * 1e3, 4),
                         "steps": n32, "dtype": "fp32 (v_mfma_f32_32x32x2_f32, exact)",
                         "frac_of_fp32_mfma_peak": round(sum(c["flops"] for c in model32._active_plan.conv_log if not c.get("onehot")) / (el32 / n32) / 1e12 / PEAK_TFLOPS["fp32"], 4)}
            del model32
            torch.cuda.empty_cache()
        # companion figure: the "x3" precision -- fp32 storage, statistics and norms, the 3x3 convolutions (ResnetBlocks,
        # stride-2 stages: 80 % of the frame's FLOP) on the bf16 matrix pipe over bf16x3 operands (engine.X3Conv): the parity bar
        # of the fp32 path at a multiple of its speed
        if args.precision == "bf16" and not face:
            try:
                _, model_x3 = build_model("x3")
                for s in range(S):
                    getattr(model_x3, "netG%d" % s).load_state_dict(getattr(model, "netG%d" % s).state_dict())
                model_x3.engine.refresh_weights()
                ex3 = errors_of(model_x3)
                nx3 = max(args.steps // 2, 5)
                elx3, _, _ = timed_fps(model_x3, nx3, 2, windows=3, min_warm_s=0.3)
                logx3 = model_x3._active_plan.conv_log
                x3_line = {"value": round(nx3 / elx3, 3), "unit": "frames/s", "ms_per_step": round(elx3 / nx3 * 1e3, 4), "steps": nx3,
                           "dtype": "fp32 storage / statistics / norms; 3x3 convolutions as bf16x3 (hi + lo operands, 3 bf16 MFMA products, fp32 accumulate)",
                           "parity": ex3, "max_rel": max(v["max_rel"] for v in ex3.values()),
                           "ok_1e-3": bool(max(v["max_rel"] for v in ex3.values()) <= 1e-3 and all(v["finite"] for v in ex3.values())),
                           "x3_convs_per_frame": sum(1 for c in logx3 if c.get("x3")), "convs_per_frame": len(logx3),
                           "x3_flop_share": round(sum(c["flops"] for c in logx3 if c.get("x3")) / max(sum(c["flops"] for c in logx3), 1.0), 3)}
                del model_x3
                torch.cuda.empty_cache()
            except Exception as ex:
                import traceback
                traceback.print_exc()
                x3_line = {"error": repr(ex)[:400]}

    # ---------------- the timed region ----------------
    elapsed, window_s, warm_frames = timed_fps(model, args.steps, args.warmup, args.windows, args.min_warmup_s)
    fps = world * args.steps / elapsed                           # world == --gpus (checked at start-up)
    fp = model._active_plan
    finite = bool(torch.isfinite(fp.out["fake_B"]).all().item())

    # ---------------- roofline of the dominant kernel (HIP events, same plan, same stream) ----
    def kernel_table(fp_, nprof):
        """HIP-event durations of every launch of an eager single-stream replay of the frame plan, joined with the plan's
        conv census: per (tile, split-K, members) configuration the summed time and algorithmic FLOP."""
        # one entry per LAUNCH: the two members of a paired launch (v2v_conv2d_pair) are one kernel
        launches, i = [], 0
        mfma_log = [c for c in fp_.conv_log if not c.get("onehot")]       # one-hot stems are gather-sums, not conv2d launches
        while i < len(mfma_log):
            c = mfma_log[i]
            if c.get("pair"):
                c2 = mfma_log[i + 1]
                launches.append(dict(c, flops=c["flops"] + c2["flops"], label=c["label"] + " + " + c2["label"], members=2))     # (keeps fused_norm)
                i += 2
            else:
                launches.append(dict(c, members=1))
                i += 1
        acc = {}
        for _ in range(nprof):
            rows = fp_.plan.profile()
            convs = [r for r in rows if r[0] == KERNEL_FAMILY]
            assert len(convs) == len(launches), (len(convs), len(launches))
            for (name, label, ms), c in zip(convs, launches):
                key = (c["tile"], c.get("splitk", 1), c["members"])
                a = acc.setdefault(key, dict(ms=0.0, flops=0.0, launches=0))
                a["ms"] += ms; a["flops"] += c["flops"]; a["launches"] += 1
        total_ms = {}
        for name, label, ms in rows:
            total_ms[name] = total_ms.get(name, 0.0) + ms
        return acc, rows, convs, launches, mfma_log, total_ms

    def tile_label(dom, fused=False):
        from vid2vid_amd.engine import TILE_CFGS, PATCH_CFGS, S2_CFGS, T2_CFGS, S7_CFGS
        if dom[0] in S7_CFGS:
            th_, tw_, bn = S7_CFGS[dom[0]]
            return "conv7x7_pp3_kernel", "%dx%d px x %d (7x7 window, single-phase schedule),splitK=%d" % (th_, tw_, bn, dom[1])
        if dom[0] in (140, 141):
            th_, tw_, bn = PATCH_CFGS[dom[0]]
            return "conv3x3_one_kernel" if dom[0] == 140 else "conv3x3_one_db_kernel", \
                   "%dx%d px x %d (persistent, weights resident%s),splitK=1" % (th_, tw_, bn, ", two patch buffers" if dom[0] == 141 else "")
        if dom[0] in T2_CFGS:
            th_, tw_, bn = T2_CFGS[dom[0]]
            return "conv3x3_t2_kernel", "%dx%d positions x %d x 4 classes (transposed stride 2),splitK=1" % (th_, tw_, bn)
        if dom[0] in S2_CFGS:
            th_, tw_, bn = S2_CFGS[dom[0]]
            return "conv3x3_s2_kernel", "%dx%d px x %d (stride 2, plane-resident patch),splitK=1" % (th_, tw_, bn)
        if dom[0] in PATCH_CFGS:
            th_, tw_, bn = PATCH_CFGS[dom[0]]
            fam = "conv3x3_pp3_kernel" if dom[0] >= 80 else "conv3x3_pp2_kernel" if dom[0] >= 70 else "conv3x3_pp_kernel" if dom[0] >= 50 else "conv3x3_patch_kernel"
            return fam, "%dx%d px x %d,splitK=%d%s%s" % (th_, tw_, bn, dom[1], ",paired launch (2 convolutions)" if dom[2] == 2 else "",
                                                           ",norm+act+residual fused" if fused else "")
        if dom[0] in (60, 61, 62):
            return {60: "conv7x7_head_kernel", 61: "conv7x7_c8_kernel", 62: "conv7x7_rowsum_kernel"}[dom[0]], "halo patch,splitK=%d" % dom[1]
        bm, bn, _ = TILE_CFGS.get(dom[0], (0, 0, False))
        return "conv_igemm_kernel", "%dx%d,splitK=%d" % (bm, bn, dom[1])

    roofline = None
    if rank == 0:
        nprof = max(args.profile_frames, 1)
        acc, rows, convs, launches, mfma_log, total_ms = kernel_table(fp, nprof)
        dom = max(acc, key=lambda k: acc[k]["flops"])
        a = acc[dom]
        ach = a["flops"] / (a["ms"] * 1e-3) / 1e12
        # the ResnetBlock layers alone (the 1024 -> 1024 3x3 convolutions, 36 per frame)
        rb = [(ms, c) for (n_, l_, ms), c in zip(convs, launches) if c["cin"] == 1024 and c["cout"] == 1024 and c["KH"] == 3]
        rb_tf = (sum(c["flops"] for _, c in rb) / (sum(ms for ms, _ in rb) * 1e-3) / 1e12) if rb else None
        peak = PEAK_TFLOPS[args.precision]
        fam, tile_name = tile_label(dom, bool(rb and rb[0][1].get("fused_norm")))
        # committed measurements of the same kernel configuration (separate rocprofv3 runs, scripts/gpu_r2.sh): HBM traffic
        # per launch (PMC passes) and the kernel's average duration INSIDE the graph (kernel trace of this bench command)
        traffic = traffic_detail = in_graph = None
        try:
            import glob
            esz = 2 if args.precision == "bf16" else 4
            c0 = rb[0][1] if rb else None
            # input + weights + output: fp32 raw (4 B) for conv + bn_apply, or with the norm fused in the activation dtype plus the
            # residual the second convolution of a ResnetBlock adds (the measured launch is that second one)
            fused_ = bool(c0 and c0.get("fused_norm"))
            alg = None if c0 is None else c0["members"] * (c0["N"] * c0["H"] * c0["W"] * c0["cin"] * esz + c0["cout"] * c0["cin"] * 9 * esz
                                                            + c0["N"] * c0["OH"] * c0["OW"] * c0["cout"] * (2 * esz if fused_ else 4))
            import re
            natural = lambda f: [int(t) if t.isdigit() else t for t in re.split(r"(\d+)", os.path.basename(f))]      # r02_a51 after r02_a7
            for fn in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")), key=natural, reverse=True):
                tj = json.load(open(fn))
                if list(tj.get("cfg", [])) == [dom[0], dom[1], dom[2]] and tj.get("hbm_bytes_per_launch"):
                    traffic = tj["hbm_bytes_per_launch"]
                    traffic_detail = {"hbm_bytes_per_launch": traffic, "algorithmic_bytes_per_launch": alg,
                                      "source": "profiles/" + os.path.basename(fn) + " (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                                      "passes, FETCH_SIZE x2 per the gfx950 correction; the 1024->1024 3x3 layer)"}
                    break
            for fn in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_in_graph.json")), key=natural, reverse=True):
                gj = json.load(open(fn))
                if list(gj.get("cfg", [])) == [dom[0], dom[1], dom[2]] and gj.get("avg_us"):
                    in_graph = {"avg_launch_us": gj["avg_us"], "achieved": round(a["flops"] / a["launches"] / gj["avg_us"] / 1e6, 2),
                                "unit": "TFLOP/s", "frac": round(a["flops"] / a["launches"] / gj["avg_us"] / 1e6 / peak, 4),
                                "source": "profiles/" + os.path.basename(fn) + " (rocprofv3 --kernel-trace --stats of this bench command: "
                                "the kernel as it runs in the timed region, lanes sharing the chip)"}
                    break
        except Exception:
            pass
        frame_flops = sum(c["flops"] for c in mfma_log)          # executed MFMA work (gather-sum stems excluded)
        # the dominant kernel AS IT RUNS IN THE TIMED REGION: device wall-clock stamps captured in front of and behind every
        # op inside the per-lane segment graphs (v2v_plan_timeline_graph), the lanes sharing the chip exactly as in a frame
        # replay; the stamps cost a dispatch (~2-4 us) per op, which is inside the measured interval -- an upper bound of
        # the kernel's duration.  This is the figure `frac` is computed from; the alone-on-chip HIP-event figure is `eager`.
        live = None
        try:
            if opt.use_graph:
                durs = []
                for _ in range(5):
                    tl = [r for r in fp.plan.timeline(graph=True) if r[0] == KERNEL_FAMILY]
                    assert len(tl) == len(launches)
                    durs += [(r[4] - r[3]) * 1e3 for r, c in zip(tl, launches) if (c["tile"], c.get("splitk", 1), c["members"]) == dom]
                if durs:
                    live = sum(durs) / len(durs)
        except Exception as ex:
            sys.stderr.write("in-graph timeline failed: %r\n" % (ex,))
        fpl = a["flops"] / a["launches"]
        eager_us = a["ms"] * 1e3 / a["launches"]
        # `frac`: the kernel as it runs IN the timed region.  First choice: the committed rocprofv3 kernel-trace average of this
        # very bench command for the same (tile, split-K, members) configuration (pure kernel time; profiles/*_in_graph.json);
        # else this run's own timeline stamps (an upper bound: two extra dispatches sit inside the interval, +6 us measured);
        # else the eager figure.  All three are in the line.
        if in_graph is not None:
            used_us, how = in_graph["avg_launch_us"], in_graph["source"]
        elif live is not None:
            used_us, how = live, ("in the frame graph: device wall-clock stamps around every launch inside the per-lane segment graphs "
                                  "(v2v_plan_timeline_graph, 5 replays; the stamps' own dispatches are inside the interval)")
        else:
            used_us, how = eager_us, "HIP events around every launch of an eager single-stream replay of the frame plan (kernel alone on the chip)"
        ach_used = fpl / used_us / 1e6
        roofline = {
            "bound": "mfma",
            "kernel": "%s<%s,%s> (tile config %d)" % (fam, "bf16" if args.precision == "bf16" else "f32", tile_name, dom[0]),
            "achieved": round(ach_used, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach_used / peak, 4),
            "traffic": traffic, "traffic_detail": traffic_detail,
            "avg_launch_us": round(used_us, 2),
            "launches_per_frame": a["launches"] // nprof,
            "flop_per_launch": fpl,
            "measured": how,
            "in_graph_live": None if live is None else {"avg_launch_us": round(live, 2), "achieved": round(fpl / live / 1e6, 2), "frac": round(fpl / live / 1e6 / peak, 4),
                                                        "measured": "device wall-clock stamps around every launch inside the per-lane segment graphs of THIS run "
                                                                    "(v2v_plan_timeline_graph, 5 replays); upper bound: the stamps' own dispatches are inside the interval"},
            "eager": {"avg_launch_us": round(eager_us, 2), "achieved": round(ach, 2), "frac": round(ach / peak, 4),
                      "measured": "HIP events around every launch of an eager single-stream replay of the frame plan (kernel alone on the chip)"},
            "in_graph_rocprof": in_graph,
            "resblock_1024_tflops": None if rb_tf is None else round(rb_tf, 2),
            "frame_in_graph": {"achieved": round(frame_flops / (elapsed / args.steps) / 1e12, 2),
                               "unit": "TFLOP/s", "frac": round(frame_flops / (elapsed / args.steps) / 1e12 / peak, 4),
                               "note": "all MFMA conv FLOP executed per frame / measured ms_per_step (norms, pooling, warp, gather-sum stems included in the time)"},
            "frame_ms_eager_events": round(sum(ms for _, _, ms in rows), 3),
            "per_kernel_ms": {k: round(v, 3) for k, v in sorted(total_ms.items(), key=lambda kv: -kv[1])},
        }
        if args.dump_ops:
            with open(args.dump_ops, "w") as f:
                it = iter(launches)
                json.dump([dict(op=n_, label=l_, ms=ms, tile=(lambda c: [c["tile"], c.get("splitk", 1), c["members"]])(next(it)) if n_ == KERNEL_FAMILY else None)
                           for n_, l_, ms in rows], f, indent=1)

    def hires_companion():
        """2048x1024, n_scales_spatial=3 (three generators, 415 M parameters, 5064 GFLOP per frame as dense convolutions):
        frames/s of inference() in the benchmarked dtype (median of 3 windows), the dominant kernel's fraction of the
        MFMA peak from HIP events, and fp32 parity of ONE frame against the CPU oracle (every head of the finest scale)."""
        nonlocal A, I, frames, lab, inst
        Hh, Wh, Sh, Lh = 1024, 2048, 3, 4
        A = I = frames = lab = inst = None               # release the 512x256 sequence

        def build_h(precision):
            o = make_opt(label_nc=35, use_instance=True, fg=True, use_real_img=True, random_init_ok=True, loadSize=Wh,
                         precision=precision, gpu_ids=[local_rank], n_scales_spatial=Sh)
            o.use_graph = not args.no_graph
            torch.manual_seed(0)
            m = create_model(o)
            with torch.no_grad():
                for si in range(Sh):
                    getattr(m, "netG%d" % si).model_final_flow[1].weight.mul_(0.1)
            return o, m
        oh, mh = build_h(args.precision)
        lab_h, inst_h, fr_h = synthetic.label2city_sequence(Lh + tG, Hh, Wh, seed=1234 + rank, device=dev)
        Ah, Ih = lab_h.view(1, Lh + tG, 1, Hh, Wh), inst_h.view(1, Lh + tG, 1, Hh, Wh)

        def step_h(m, t):
            k = t % Lh
            return m.inference(Ah[:, k:k + tG], fr_h[:, :tG - 1] if t == 0 else None, Ih[:, k:k + tG])
        mh.fake_B_prev = None
        tb = time.perf_counter()
        step_h(mh, 0)                                    # plan build (tile selections replayed from the cache when present)
        torch.cuda.synchronize(dev)
        build_s = time.perf_counter() - tb
        n_h = max(args.steps // 2, 10)
        t, tw = 1, time.perf_counter()
        while t < 3 or time.perf_counter() - tw < args.min_warmup_s:
            step_h(mh, t); t += 1
            torch.cuda.synchronize(dev)
        els = []
        for _ in range(3):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _k in range(n_h):
                step_h(mh, t); t += 1
            torch.cuda.synchronize(dev)
            els.append(time.perf_counter() - t0)
        el = sorted(els)[1]
        fph = mh._active_plan
        acc, rows, convs, launches, mfma_log, total_ms = kernel_table(fph, 1)
        peak = PEAK_TFLOPS[args.precision]
        if args.dump_ops:                                # per-op table of the 2048x1024 frame (scripts/per_layer_roofline.py reads it)
            it = iter(launches)
            with open(args.dump_ops + ".hires.json", "w") as f:
                json.dump([dict(op=n_, label=l_, ms=ms, **({k: c_[k] for k in ("tile", "splitk", "members", "flops", "cin", "cout", "KH", "H", "W", "OH", "OW", "stride") if k in c_}
                                                           if n_ == KERNEL_FAMILY and (c_ := next(it)) is not None else {}))
                           for n_, l_, ms in rows], f, indent=1)
        # the kernel configuration that holds the most TIME of this frame (at 2048x1024 the FLOP-heaviest one, the paired
        # 1024-channel launch, is a tenth of the frame): priced per launch against max(FLOP / MFMA peak, bytes / HBM peak) with
        # bytes = input + output + weights once at the storage size
        esz_h = 2 if args.precision == "bf16" else 4
        HBM_PEAK = 8000.0                                  # GB/s, MI355X_MICROARCH.md
        per_cfg = {}
        for (n_, l_, ms), c in zip(convs, launches):
            key = (c["tile"], c.get("splitk", 1), c["members"])
            mem = c["members"]
            by = mem * ((c["N"] * c["H"] * c["W"] * c["cin"] + c["N"] * c["OH"] * c["OW"] * c["cout"]
                         + c["cout"] * c["cin"] * c["KH"] * c.get("KW", c["KH"])) * esz_h)
            d = per_cfg.setdefault(key, dict(ms=0.0, flops=0.0, bytes=0.0, launches=0, t_mfma=0.0, t_hbm=0.0))
            d["ms"] += ms; d["flops"] += c["flops"]; d["bytes"] += by; d["launches"] += 1
            d["t_mfma"] += c["flops"] / (peak * 1e12) * 1e3; d["t_hbm"] += by / (HBM_PEAK * 1e9) * 1e3
        dom = max(per_cfg, key=lambda k: per_cfg[k]["ms"])
        a = per_cfg[dom]
        fam, tname = tile_label(dom)
        hbm_bound = a["t_hbm"] >= a["t_mfma"]
        ach_h = (a["bytes"] / (a["ms"] * 1e-3) / 1e9) if hbm_bound else (a["flops"] / (a["ms"] * 1e-3) / 1e12)
        pk_h = HBM_PEAK if hbm_bound else peak
        domf = max(acc, key=lambda k: acc[k]["flops"])
        af = acc[domf]
        by_time = sorted(acc.items(), key=lambda kv: -kv[1]["ms"])[:4]
        sum_bound = sum(max(v["t_mfma"], v["t_hbm"]) for v in per_cfg.values())
        line = {"metric": "synthesized frames/sec (Vid2VidModelG.inference, %dx%d)" % (Wh, Hh),
                "value": round(n_h / el, 3), "unit": "frames/s", "ms_per_step": round(el / n_h * 1e3, 3), "steps": n_h,
                "windows_ms_per_step": [round(e / n_h * 1e3, 3) for e in els], "dtype": args.precision,
                "workload": "label2city %dx%d inference, n_scales_spatial=%d, --fg --use_instance, ngf=128 (%.1fM params random-init, %.0f GFLOP/frame "
                            "as dense convolutions), batch 1, sequence resident in HBM" % (Wh, Hh, Sh, sum(q.numel() for q in mh.parameters()) / 1e6,
                                                                                        sum(c["flops"] for c in fph.conv_log) / 1e9),
                "launches_per_frame": fph.plan.num_ops, "plan_build_s": round(build_s, 1),
                "roofline": {"bound": "hbm" if hbm_bound else "mfma",
                             "kernel": "%s<%s,%s> (tile config %d): the configuration holding the most time of this frame" % (fam, args.precision, tname, dom[0]),
                             "achieved": round(ach_h, 2), "peak": pk_h, "unit": "GB/s" if hbm_bound else "TFLOP/s",
                             "frac": round(ach_h / pk_h, 4), "launches_per_frame": a["launches"],
                             "avg_launch_us": round(a["ms"] * 1e3 / a["launches"], 2), "ms_per_frame": round(a["ms"], 3),
                             "algorithmic_bytes": a["bytes"], "algorithmic_flop": a["flops"],
                             "measured": "HIP events around every launch of an eager replay (alone on the chip)",
                             "flop_heaviest": {"kernel": "%s<%s> (tile config %d)" % (tile_label(domf)[0], tile_label(domf)[1], domf[0]),
                                               "achieved": round(af["flops"] / (af["ms"] * 1e-3) / 1e12, 2), "unit": "TFLOP/s",
                                               "frac": round(af["flops"] / (af["ms"] * 1e-3) / 1e12 / peak, 4),
                                               "ms_per_frame": round(af["ms"], 3), "launches_per_frame": af["launches"]},
                             "frame_vs_per_layer_bounds": {"sum_bounds_ms": round(sum_bound, 3), "frame_ms": round(el / n_h * 1e3, 3),
                                                           "frac": round(sum_bound / (el / n_h * 1e3), 4),
                                                           "note": "sum over the conv launches of max(FLOP / %.0f TFLOP/s, bytes / %.0f GB/s) / measured frame time" % (peak, HBM_PEAK)},
                             "frame_in_graph": {"achieved": round(sum(c["flops"] for c in mfma_log) / (el / n_h) / 1e12, 2), "unit": "TFLOP/s",
                                                "frac": round(sum(c["flops"] for c in mfma_log) / (el / n_h) / 1e12 / peak, 4)},
                             "frame_ms_eager_events": round(sum(ms for _, _, ms in rows), 3),
                             "slowest_configs_ms": {"tile %d/S%d/x%d" % k: round(v["ms"], 3) for k, v in by_time},
                             "per_kernel_ms": {k: round(v, 3) for k, v in sorted(total_ms.items(), key=lambda kv: -kv[1])}}}
        if do_cpu:
            from oracle import vid2vid_oracle as O
            sds = [{k: v.detach().float().cpu() for k, v in getattr(mh, "netG%d" % s_).state_dict().items()} for s_ in range(Sh)]
            orc = O.InferenceOracle(sds, 35, True, True, [26], oh.n_downsample_G, oh.n_blocks, oh.n_blocks_local)
            c0 = time.perf_counter()
            ref_f, _ = orc.step(lab_h[:tG].cpu().view(1, tG, 1, Hh, Wh), fr_h[:, :tG - 1].cpu(), inst_h[:tG].cpu().view(1, tG, 1, Hh, Wh))
            cpu_s = time.perf_counter() - c0
            refs_h = dict(fake_B=ref_f, raw=orc.last["raw0"], flow=orc.last["flow0"], weight=orc.last["weight0"])

            def errs(m):
                m.fake_B_prev = None
                fake, _ = step_h(m, 0)
                fpm = m._active_plan
                got = dict(fake_B=fake, raw=fpm.out["raw0"], flow=fpm.out["flow0"], weight=fpm.out["weight0"])
                res = {}
                for k, r in refs_h.items():
                    g = got[k].detach().float().cpu()
                    e = (g - r).abs() / (r.abs() + r.pow(2).mean().sqrt().item() + 1e-12)
                    res[k] = {"max_rel": float("%.3e" % e.max().item()), "mean_rel": float("%.3e" % e.mean().item()),
                              "finite": bool(torch.isfinite(g).all().item())}
                return res
            e16 = errs(mh) if args.precision == "bf16" else None
            sd_keep = [getattr(mh, "netG%d" % s_).state_dict() for s_ in range(Sh)]
            del mh
            torch.cuda.empty_cache()
            _, m32 = build_h("fp32")
            for s_ in range(Sh):
                getattr(m32, "netG%d" % s_).load_state_dict(sd_keep[s_])
            m32.engine.refresh_weights()
            e32 = errs(m32)
            line["parity"] = {"frames": 1, "reference": "oracle/vid2vid_oracle.py, first generated frame from the given real frames, all 3 scales",
                              "tolerance_fp32": 1e-3, "fp32": e32, "fp32_max_rel": max(v["max_rel"] for v in e32.values()),
                              "fp32_ok": bool(max(v["max_rel"] for v in e32.values()) <= 1e-3 and all(v["finite"] for v in e32.values())),
                              "bf16": e16}
            line["cpu_baseline"] = {"value": round(1.0 / cpu_s, 4), "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
                                    "sample": "1 frame of the same %dx%d / 3-scale workload (no warm-up), fp32, oracle/vid2vid_oracle.py" % (Wh, Hh)}
            del m32
            torch.cuda.empty_cache()
            # the parity-holding THROUGHPUT mode at this resolution: fp32 storage / statistics / norms, the 3x3
            # convolutions as three bf16 MFMA products (engine.X3Conv) -- one oracle-checked frame, then frames/s
            try:
                _, mx = build_h("x3")
                for s_ in range(Sh):
                    getattr(mx, "netG%d" % s_).load_state_dict(sd_keep[s_])
                mx.engine.refresh_weights()
                ex = errs(mx)
                for t_ in range(1, 3):
                    step_h(mx, t_)
                torch.cuda.synchronize(dev)
                n_x = 8
                t0 = time.perf_counter()
                for t_ in range(3, 3 + n_x):
                    step_h(mx, t_)
                torch.cuda.synchronize(dev)
                el_x = time.perf_counter() - t0
                mxr = max(v["max_rel"] for v in ex.values())
                line["x3"] = {"value": round(n_x / el_x, 3), "unit": "frames/s", "ms_per_step": round(el_x / n_x * 1e3, 3), "steps": n_x,
                              "max_rel": mxr, "ok_1e-3": bool(mxr <= 1e-3 and all(v["finite"] for v in ex.values())), "errors": ex,
                              "dtype": "x3 (fp32 storage / norms, 3x3 convolutions as three bf16 MFMA products)"}
                del mx
                torch.cuda.empty_cache()
            except Exception as ex_:
                line["x3"] = {"error": repr(ex_)[:300]}
        return line

    def c1_clip(c4=False):
        """BASELINE configs[0] literally: label2city 256x128, n_scales_spatial=1, a 2-frame clip (two generated frames behind
        the tG-1 given real frames; the reference's --use_single_G nets exist only for loadSize 512 / 1024 / 2048),
        full width (ngf 128, 9 blocks, --fg --use_instance).  The CPU oracle generates the clip (timed: the `cpu_baseline` of
        THIS config), the fp32 and x3 paths are gated against it per pixel at 1e-3 on every head, and the clip is timed on
        the GPU in the benchmarked dtype (sequence reset + first-frame pyramid + 2 frames per clip).
        c4=True: BASELINE configs[3] instead (edge2face 512x512: 15 raw input maps per frame, no instance map, no fg tower,
        scripts/face/test_512.sh) -- the same two oracle-checked frames, and the throughput of a RUNNING sequence (steady
        state, no reset per clip) since that config's metric is frames of inference()."""
        from oracle import vid2vid_oracle as O
        Hc, Wc, nfc = (512, 512, 2) if c4 else (128, 256, 2)

        def build_c(precision):
            if c4:
                o = make_opt(label_nc=0, input_nc=15, use_instance=False, fg=False, use_real_img=True, random_init_ok=True,
                             dataroot="datasets/face/", loadSize=Wc, precision=precision, gpu_ids=[local_rank], n_scales_spatial=1)
            else:
                o = make_opt(label_nc=35, use_instance=True, fg=True, use_real_img=True, random_init_ok=True, loadSize=Wc,
                             precision=precision, gpu_ids=[local_rank], n_scales_spatial=1)
            o.use_graph = not args.no_graph
            o.frame_tune = 0                         # a side figure: per-shape tile search only
            torch.manual_seed(0)
            m = create_model(o)
            with torch.no_grad():
                m.netG0.model_final_flow[1].weight.mul_(0.1)
            return o, m
        oc, mc = build_c(args.precision)
        if c4:
            Ac, fr_c = synthetic.edge2face_sequence(nfc + tG - 1 + 14, Hc, Wc, seed=1234, device=dev)
            Ic = None
        else:
            lab_c, inst_c, fr_c = synthetic.label2city_sequence(nfc + tG - 1, Hc, Wc, seed=1234, device=dev)
            Ac, Ic = lab_c.view(1, -1, 1, Hc, Wc), inst_c.view(1, -1, 1, Hc, Wc)

        def step_c(m, t, k=None):
            k = t if k is None else k
            return m.inference(Ac[:, k:k + tG], fr_c[:, :tG - 1] if t == 0 else None, None if Ic is None else Ic[:, k:k + tG])

        def clip(m):
            m.fake_B_prev = None
            return [step_c(m, t)[0] for t in range(nfc)]
        clip(mc); clip(mc)
        torch.cuda.synchronize(dev)
        if c4:                                        # a running sequence, the resident maps cycled: frames of inference() per second
            ncl, nrun = 1, 40
            for t in range(2, 10):
                step_c(mc, t, t % 14)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for t in range(10, 10 + nrun):
                step_c(mc, t, t % 14)
            torch.cuda.synchronize(dev)
            el = time.perf_counter() - t0
        else:
            ncl = 20
            t0 = time.perf_counter()
            for _ in range(ncl):
                clip(mc)
            torch.cuda.synchronize(dev)
            el = time.perf_counter() - t0
        sd = {k: v.detach().float().cpu() for k, v in mc.netG0.state_dict().items()}
        if c4:
            orc = O.InferenceOracle([sd], 0, False, False, [], oc.n_downsample_G, oc.n_blocks, oc.n_blocks_local)
            acc, fcc = Ac.cpu(), fr_c.cpu()
        else:
            orc = O.InferenceOracle([sd], 35, True, True, [26], oc.n_downsample_G, oc.n_blocks, oc.n_blocks_local)
            lcc, icc, fcc = lab_c.cpu(), inst_c.cpu(), fr_c.cpu()
        refs_c, prevs_c, cpu_s = [], [], None
        for rep in range(2):                          # the clip twice: the second run is the timed one (warm allocator / threads)
            orc.fake_B_prev = None
            refs_c, prevs_c = [], []
            c0 = time.perf_counter()
            for t in range(nfc):
                prevs_c.append(None if orc.fake_B_prev is None else [q.clone() for q in orc.fake_B_prev])
                if c4:
                    f_, _ = orc.step(acc[:, t:t + tG], fcc[:, :tG - 1] if t == 0 else None, None)
                else:
                    f_, _ = orc.step(lcc[t:t + tG].view(1, tG, 1, Hc, Wc), fcc[:, :tG - 1] if t == 0 else None, icc[t:t + tG].view(1, tG, 1, Hc, Wc))
                refs_c.append(dict(fake_B=f_.clone(), raw=orc.last["raw0"].clone(), flow=orc.last["flow0"].clone(), weight=orc.last["weight0"].clone()))
            cpu_s = time.perf_counter() - c0

        def errs(m):
            m.fake_B_prev = None
            worst = {}
            for t in range(nfc):
                if t > 0:
                    m._active_plan.prev[0].copy_(prevs_c[t][0])
                fake, _ = step_c(m, t)
                fpm = m._active_plan
                got = dict(fake_B=fake, raw=fpm.out["raw0"], flow=fpm.out["flow0"], weight=fpm.out["weight0"])
                for k, r in refs_c[t].items():
                    g = got[k].detach().float().cpu()
                    e = (g - r).abs() / (r.abs() + r.pow(2).mean().sqrt().item() + 1e-12)
                    w = worst.setdefault(k, {"max_rel": 0.0, "mean_rel": 0.0, "finite": True})
                    w["max_rel"] = max(w["max_rel"], e.max().item()); w["mean_rel"] = max(w["mean_rel"], e.mean().item())
                    w["finite"] = w["finite"] and bool(torch.isfinite(g).all().item())
            return {k: {"max_rel": float("%.3e" % v["max_rel"]), "mean_rel": float("%.3e" % v["mean_rel"]), "finite": v["finite"]} for k, v in worst.items()}
        par = {"frames": nfc, "tolerance_fp32": 1e-3, "measure": "per pixel |got-ref| / (|ref| + rms(ref)), every frame from the oracle's own previous frames"}
        par[args.precision] = errs(mc)
        sd_keep = mc.netG0.state_dict()
        for prec in ("fp32", "x3"):
            if prec == args.precision:
                continue
            _, m2 = build_c(prec)
            m2.netG0.load_state_dict(sd_keep)
            m2.engine.refresh_weights()
            par[prec] = errs(m2)
            del m2
            torch.cuda.empty_cache()
        par["fp32_max_rel"] = max(v["max_rel"] for v in par["fp32"].values())
        par["x3_max_rel"] = max(v["max_rel"] for v in par["x3"].values())
        par["fp32_ok"] = bool(par["fp32_max_rel"] <= 1e-3 and all(v["finite"] for v in par["fp32"].values()))
        par["x3_ok"] = bool(par["x3_max_rel"] <= 1e-3 and all(v["finite"] for v in par["x3"].values()))
        cpu_line = {"value": round(nfc / cpu_s, 4), "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
                    "sample": "the same %d-frame clip (second of two runs), fp32, oracle/vid2vid_oracle.py; clip %.2f s" % (nfc, cpu_s)}
        if c4:
            return {"workload": "BASELINE configs[3]: edge2face %dx%d, input_nc=15 (45 generator input channels), no instance map, no fg tower, "
                                "n_scales_spatial=1, ngf=128 n_blocks=9 (%.1fM params random-init), batch 1, maps resident in HBM"
                                % (Wc, Hc, sum(q.numel() for q in mc.parameters()) / 1e6),
                    "metric": "synthesized frames/sec (Vid2VidModelG.inference, %dx%d edge2face)" % (Wc, Hc),
                    "value": round(nrun / el, 3), "unit": "frames/s", "ms_per_step": round(el / nrun * 1e3, 4), "steps": nrun, "dtype": args.precision,
                    "note": "running sequence (no reset), per-shape tile selection only; parity: the first %d frames vs the CPU oracle" % nfc,
                    "parity": par, "cpu_baseline": cpu_line}
        return {"workload": "BASELINE configs[0]: label2city %dx%d, n_scales_spatial=1, %d-frame clip from %d given real frames (--use_real_img), "
                            "--fg --use_instance, ngf=128 n_blocks=9" % (Wc, Hc, nfc, tG - 1),
                "value": round(ncl * nfc / el, 3), "unit": "frames/s", "ms_per_clip": round(el / ncl * 1e3, 3), "clips": ncl, "dtype": args.precision,
                "note": "every clip resets the sequence (first-frame pyramid from the real frames) and generates %d frames" % nfc,
                "parity": par, "cpu_baseline": cpu_line}

    sys.stdout = _stdout
    leg_s = {"headline": round(time.perf_counter() - t_main, 1)}      # wall seconds per leg of the default command (full record)
    if rank == 0:
        out = {
            "metric": "synthesized frames/sec (Vid2VidModelG.inference, %dx%d)" % (W, H),
            "value": round(fps, 3), "unit": "frames/s", "n_gpus": args.gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.precision, "data": "synthetic",
            "timing": {"windows_ms_per_step": [round(e / args.steps * 1e3, 4) for e in window_s], "statistic": "median window",
                       "warmup_frames_run": warm_frames, "min_warmup_s": args.min_warmup_s,
                       "note": "%d windows of exactly --steps frames, each bracketed by barrier + synchronize and MAX-reduced over ranks; "
                               "`value` / `ms_per_step` are the median window" % len(window_s)},
            "config": {"workload": "%s %dx%d inference, n_scales_spatial=%d, %s, ngf=128 n_blocks=9 "
                                   "(%.1fM params random-init, %.0f GFLOP/frame as dense convolutions%s), batch 1 per sequence, 1 sequence per GPU"
                                   % (args.dataset, W, H, args.scales, "input_nc=15, no fg tower" if face else "--fg --use_instance", sum(q.numel() for q in model.parameters()) / 1e6,
                                      sum(c["flops"] for c in fp.conv_log) / 1e9,
                                      (", of which the %d one-hot label stems (%.0f GFLOP dense) run as exact weight gather-sums"
                                       % (sum(1 for c in fp.conv_log if c.get("onehot")), sum(c["flops"] for c in fp.conv_log if c.get("onehot")) / 1e9))
                                      if any(c.get("onehot") for c in fp.conv_log) else ""),
                       "launches_per_frame": fp.plan.num_ops, "hipgraph": bool(opt.use_graph),
                       "graph_lanes": 3 if getattr(fp, "lanes", False) else 1,
                       "paired_launches": bool(getattr(fp, "twin", False)),
                       "tile_selection": tune_src or ("V2V_TUNE_CACHE" if os.environ.get("V2V_TUNE_CACHE") and not args.retune else "measured in this run"),
                       "frame_tune": getattr(fp, "frame_tune_log", None),
                       "world_size": world, "backend": (dist.get_backend() if world > 1 else None),
                       "parallelism": "replicas x%d (independent sequences, no collective)" % args.gpus,
                       "output_finite": finite},
            "parity": parity,
            "fp32": fp32_line,
            "x3": x3_line,
            "host_fed": host_fed,
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        # ---- companion figure: the OTHER resolution of BASELINE's metric, 2048x1024 with n_scales_spatial=3 (configs[4] geometry) ----
        if world == 1 and not args.no_hires and not face and args.scales == 1 and (W, H) == (512, 256):
            sys.stdout = sys.stderr
            try:
                model = fp = None
                torch.cuda.empty_cache()
                _t = time.perf_counter()
                out["hires"] = hires_companion()
                leg_s["hires"] = round(time.perf_counter() - _t, 1)
            except Exception as ex:
                import traceback
                traceback.print_exc()
                out["hires"] = {"error": repr(ex)[:400]}
            finally:
                sys.stdout = _stdout
        # ---- BASELINE configs[0] literally: the 256x128 2-frame clip, GPU vs the CPU oracle (parity + both timings) ----
        if world == 1 and do_cpu and not args.no_c1 and not face and args.scales == 1 and (W, H) == (512, 256):
            sys.stdout = sys.stderr
            try:
                model = fp = None
                torch.cuda.empty_cache()
                _t = time.perf_counter()
                out["c1"] = c1_clip()
                leg_s["c1"] = round(time.perf_counter() - _t, 1)
            except Exception as ex:
                import traceback
                traceback.print_exc()
                out["c1"] = {"error": repr(ex)[:400]}
            finally:
                sys.stdout = _stdout
        # ---- BASELINE configs[3]: edge2face 512x512 (two oracle-checked frames, steady-state frames/s) ----
        if world == 1 and do_cpu and not args.no_c4 and not face and args.scales == 1 and (W, H) == (512, 256):
            sys.stdout = sys.stderr
            try:
                model = fp = None
                torch.cuda.empty_cache()
                _t = time.perf_counter()
                out["c4"] = c1_clip(c4=True)
                leg_s["c4"] = round(time.perf_counter() - _t, 1)
            except Exception as ex:
                import traceback
                traceback.print_exc()
                out["c4"] = {"error": repr(ex)[:400]}
            finally:
                sys.stdout = _stdout
        # ---- companion figure: the training step (train.py inner loop) on the same geometry, a short run ----
        import argparse

        def train_companion(key, **over):
            try:
                torch.cuda.empty_cache()
                targs = argparse.Namespace(**vars(args))
                targs.mode, targs.steps, targs.warmup, targs.no_vgg = "train", 6, 2, False
                for k_, v_ in over.items():
                    setattr(targs, k_, v_)
                _t = time.perf_counter()
                tr = run_train(targs, dev, rank, 1, local_rank, emit=False)
                leg_s[key] = round(time.perf_counter() - _t, 1)
                out[key] = {"metric": tr["metric"], "value": tr["value"], "unit": tr["unit"], "ms_per_step": tr["ms_per_step"],
                            "steps": tr["steps"], "warmup": tr["warmup"], "dtype": tr["dtype"],
                            "frac_of_mfma_peak": tr["roofline"]["frac"], "tflops": tr["roofline"]["achieved"],
                            "dominant": tr["roofline"].get("dominant"),
                            "peak_memory_gb": tr["config"].get("peak_memory_gb"),
                            "workload": tr["config"]["workload"], "output_finite": tr["config"]["output_finite"],
                            "parity": tr["parity"], "flownet2": tr["flownet2"],
                            "note": "python bench.py --mode train%s: the full line (per-kind FLOP, launches per step)"
                                    % "".join(" --%s %s" % (k_.replace("_", "-"), v_) for k_, v_ in over.items() if k_ in ("width", "height", "scales", "num_D", "frames_per_gpu"))}
            except Exception as ex:             # the headline line must not depend on the companion run
                import traceback
                traceback.print_exc()
                out[key] = {"error": repr(ex)[:300]}
            finally:
                sys.stdout = _stdout
        if world == 1 and not args.no_train_line and not face and args.scales == 1:
            model = None
            train_companion("train")
        # ---- BASELINE configs[4] geometry as a TRAINING chunk on one GPU: 2048x1024, n_scales_spatial=3, num_D=4 + VGG as
        # scripts/street/train_2048.sh, n_frames_total=6, one frame per chunk (each generator GPU's share of the n_gpus_gen split),
        # all scales trained; fp32 chunk-vs-oracle parity (outputs, every loss, complete G / D gradients) before timing ----
        if world == 1 and not args.no_train_hires and not face and args.scales == 1 and (W, H) == (512, 256):
            model = None
            train_companion("train_hires", width=2048, height=1024, scales=3, num_D=4, frames_total=6, frames_per_gpu=1,
                            no_train_parity=bool(not args.train_hires_parity or args.no_cpu_baseline), no_bf16_train_parity=True)
        # ---- BASELINE configs[2] geometry as a training chunk on one GPU (1024x512, n_scales_spatial=2, num_D=3): frames trained/s only --
        # its fp32 chunk-vs-oracle parity is tests/test_gpu_golden.py::test_full_width_training_chunk_1024x512_s2_vs_oracle ----
        if world == 1 and not args.no_train_c3 and not face and args.scales == 1 and (W, H) == (512, 256):
            model = None
            train_companion("train_c3", width=1024, height=512, scales=2, num_D=3, frames_total=6, frames_per_gpu=2, no_train_parity=True)
        # ---- flat scalars ahead of the nested objects: the figures that carry north_star's 1e-3 parity and the second resolution ----
        flat = {}
        if x3_line and "value" in x3_line:
            flat.update(parity_value=x3_line["value"], parity_dtype="x3 (fp32 storage / norms, convolutions as three bf16 MFMA products)",
                        parity_max_rel=x3_line["max_rel"], parity_ok=x3_line["ok_1e-3"])
        if fp32_line:
            flat.update(fp32_value=fp32_line["value"], fp32_max_rel=None if parity is None else parity["fp32_max_rel"])
        if parity is not None:
            flat.update(value_max_rel=parity.get("bf16_max_rel"), value_mean_rel=parity.get("bf16_mean_rel"))
        for key, name in (("hires", "hires_value"), ("train", "train_value"), ("train_hires", "train_hires_value"), ("train_c3", "train_c3_value"),
                          ("c1", "c1_value"), ("c4", "c4_value")):
            if isinstance(out.get(key), dict) and "value" in out[key]:
                flat[name] = out[key]["value"]
        hx = out.get("hires", {}).get("x3") if isinstance(out.get("hires"), dict) else None
        if isinstance(hx, dict) and "value" in hx:                 # 2048x1024 in the mode that carries north_star's 1e-3
            flat.update(hires_x3_value=hx["value"], hires_x3_max_rel=hx["max_rel"], hires_x3_ok=hx["ok_1e-3"])
        if isinstance(out.get("train_hires"), dict) and isinstance(out["train_hires"].get("parity"), dict):
            flat["train_hires_fp32_ok"] = out["train_hires"]["parity"].get("fp32_ok")
        if isinstance(out.get("train"), dict) and isinstance(out["train"].get("parity"), dict):
            flat["train_fp32_ok"] = out["train"]["parity"].get("fp32_ok")
        leg_s["total"] = round(time.perf_counter() - t_main, 1)
        out["leg_seconds"] = leg_s
        head = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
        head.update(flat)
        head.update({k: v for k, v in out.items() if k not in head})
        out = head
        emit_record(out, _stdout)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
