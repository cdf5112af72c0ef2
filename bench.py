#!/usr/bin/env python3
"""Headline benchmark: synthesized frames/sec of Vid2VidModelG.inference() on MI355X.

Workload (BASELINE.json configs[1]): label2city 512x256, n_scales_spatial=1, --fg --use_instance,
ngf=128 / 9 blocks (411 M parameters, 2115 GFLOP per frame), batch 1 per sequence, one sequence
per GPU, bf16 storage + fp32 MFMA accumulate, random-init weights, seeded synthetic label /
instance / image sequences already resident in HBM.  A "step" is one generated frame: refresh
the plan's input buffers (device-to-device) + one hipGraph launch of the whole frame.

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

PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3}   # dense MFMA peaks, MI355X_MICROARCH.md
KERNEL_FAMILY = "conv_igemm"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
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
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--profile-frames", type=int, default=3)
    ap.add_argument("--dump-ops", default="", help="write the per-op timing table (json) here")
    return ap.parse_args()


def run_train(args, dev, rank, world, local_rank):
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
    opt = make_opt(isTrain=True, label_nc=35, use_instance=True, fg=True, random_init_ok=True, loadSize=W,
                   precision=args.precision, gpu_ids=[local_rank], n_scales_spatial=S, num_D=args.num_D,
                   n_frames_total=args.frames_total, max_frames_per_gpu=args.frames_per_gpu,
                   no_vgg=args.no_vgg, niter_fix_global=0)
    _stdout = sys.stdout
    sys.stdout = sys.stderr
    models = create_model(opt)
    modelG, modelD, flowNet, optimizer_G, optimizer_D, optimizer_D_T = create_optimizer(opt, models)
    with torch.no_grad():
        for si in range(S):
            getattr(modelG.module, "netG%d" % si).model_final_flow[1].weight.mul_(0.1)
    if world > 1:
        parallel.sync_optimizers([optimizer_G, optimizer_D] + list(optimizer_D_T))
    eng = modelG.module.engine
    tG, tD, t_scales = opt.n_frames_G, opt.n_frames_D, opt.n_scales_temporal
    n_frames_total, n_frames_load = opt.n_frames_total, modelG.module.n_frames_load
    n_seq_frames = n_frames_total + tG - 1
    lab, inst, frames = synthetic.label2city_sequence(n_seq_frames, H, W, seed=1234 + rank, device=dev)
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

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    # tile search on the first sequence (every conv shape of the step: forward, backward-data), then plain warm-up
    eng.autotune = not args.no_autotune
    t_tune = time.perf_counter()
    for _ in range(chunks_per_seq):
        step()
    eng.autotune = False
    barrier()
    t_tune = time.perf_counter() - t_tune
    for _ in range(args.warmup):
        step()
    eng.conv_log = []
    fn_flops0, fn_convs0 = flowNet.module.flops_launched, flowNet.module.convs_launched
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = tt.item()
    fps = args.gpus * args.steps * n_frames_load / elapsed
    loss_G, loss_D, t_act = state["loss"]
    finite = bool(torch.isfinite(loss_G).all().item() and torch.isfinite(loss_D).all().item())

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
            "flop_per_step": flop_step, "conv_launches_per_step": n_launch // args.steps,
            "gflop_per_step_by_kind": {k: round(v / args.steps / 1e9, 1) for k, v in sorted(by_kind.items())},
            "note": "algorithmic conv FLOP of one chunk / wall time of one chunk (host launch time included: the "
                    "training step is an eager autograd graph of v2v custom ops, not a hipGraph)",
        }
        out = {
            "metric": "frames trained/sec (train.py inner loop, %dx%d)" % (W, H),
            "value": round(fps, 3), "unit": "frames/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": "label2city %dx%d train, n_scales_spatial=%d num_D=%d n_scales_temporal=%d, --fg --use_instance%s, "
                                   "n_frames_total=%d, %d frames per chunk, niter_fix_global=0 (all scales train), G %.1fM + D %.1fM "
                                   "params random-init, FlowNet2 random-init, 1 sequence per GPU"
                                   % (W, H, S, opt.num_D, t_scales, " --no_vgg" if args.no_vgg else "", n_frames_total,
                                      n_frames_load, sum(q.numel() for q in modelG.module.parameters()) / 1e6,
                                      sum(q.numel() for q in modelD.module.parameters()) / 1e6),
                       "frames_per_step": n_frames_load, "active_temporal_scales_last_step": int(t_act),
                       "autotune_s": round(t_tune, 1),
                       "parallelism": "dp%d over sequences (RCCL all-reduce of flat gradients per optimizer)" % args.gpus,
                       "loss_G": round(float(loss_G), 4), "loss_D": round(float(loss_D), 4), "output_finite": finite},
            "roofline": roofline, "cpu_baseline": None,
        }
        print(json.dumps(out))
        sys.stdout.flush()


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)
    assert world == args.gpus or world == 1, "--gpus must match the launched world size"
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
    # N > 1: rank 0 runs the tile searches, the other ranks replay its selections (vid2vid_amd/parallel.py)
    from vid2vid_amd import parallel
    release_tuning = parallel.shared_tuning_cache(rank, world)
    face = args.dataset == "edge2face"
    if face:      # scripts/face/test_512.sh geometry: 15 raw input maps per frame, no instance map, no fg tower
        opt = make_opt(label_nc=0, input_nc=15, use_instance=False, fg=False, use_real_img=True, random_init_ok=True,
                       dataroot="datasets/face/", loadSize=W, precision=args.precision, gpu_ids=[local_rank],
                       n_scales_spatial=args.scales)
    else:
        opt = make_opt(label_nc=35, use_instance=True, fg=True, use_real_img=True, random_init_ok=True,
                       loadSize=W, precision=args.precision, gpu_ids=[local_rank], n_scales_spatial=args.scales)
    opt.use_graph = not args.no_graph
    sys.stdout.flush()
    _stdout = sys.stdout
    sys.stdout = sys.stderr                      # keep stdout for the single JSON line
    model = create_model(opt)
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

    def step(t):
        k = t % L
        model.inference(A[:, k:k + tG], frames[:, :tG - 1] if t == 0 else None, None if face else I[:, k:k + tG])

    model.fake_B_prev = None
    step(0)                                      # builds the frame plan (tile searches, graph): never inside the timed region
    torch.cuda.synchronize(dev)
    release_tuning()                             # N > 1, rank 0: the other ranks may now build with its selections
    model.fake_B_prev = None
    for t in range(args.warmup):
        step(t)

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    barrier()
    t0 = time.perf_counter()
    for t in range(args.warmup, args.warmup + args.steps):
        step(t)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = tt.item()
    fps = args.gpus * args.steps / elapsed
    fp = model._active_plan
    finite = bool(torch.isfinite(fp.out["fake_B"]).all().item())

    # ---------------- roofline of the dominant kernel (HIP events, same plan, same stream) ----
    roofline = None
    if rank == 0:
        acc = {}
        nprof = max(args.profile_frames, 1)
        for _ in range(nprof):
            rows = fp.plan.profile()
            convs = [r for r in rows if r[0] == KERNEL_FAMILY]
            assert len(convs) == len(fp.conv_log)
            for (name, label, ms), c in zip(convs, fp.conv_log):
                key = (c["tile"], c.get("splitk", 1), c.get("prefetch", 0))
                a = acc.setdefault(key, dict(ms=0.0, flops=0.0, launches=0))
                a["ms"] += ms; a["flops"] += c["flops"]; a["launches"] += 1
        total_ms = {}
        for name, label, ms in rows:
            total_ms[name] = total_ms.get(name, 0.0) + ms
        dom_tile = max(acc, key=lambda k: acc[k]["flops"])
        a = acc[dom_tile]
        ach = a["flops"] / (a["ms"] * 1e-3) / 1e12
        # the ResnetBlock layer alone (36 launches/frame of the same template instance)
        rb = [(ms, c) for (n_, l_, ms), c in zip(convs, fp.conv_log)
              if c["cin"] == 1024 and c["cout"] == 1024 and c["KH"] == 3]
        rb_tf = (sum(c["flops"] for _, c in rb) / (sum(ms for ms, _ in rb) * 1e-3) / 1e12) if rb else None
        peak = PEAK_TFLOPS[args.precision]
        from vid2vid_amd.engine import TILE_CFGS, PATCH_CFGS
        if dom_tile[0] in PATCH_CFGS:
            th_, tw_, bn = PATCH_CFGS[dom_tile[0]]
            fam = "conv3x3_pp_kernel" if dom_tile[0] >= 50 else "conv3x3_patch_kernel"
            tile_name = "%dx%d px x %d,splitK=%d" % (th_, tw_, bn, dom_tile[1])
        else:
            bm, bn, _ = TILE_CFGS.get(dom_tile[0], (0, 0, False))
            fam = "conv_igemm_kernel"
            tile_name = "%dx%d,splitK=%d,prefetch=%d" % (bm, bn, dom_tile[1], dom_tile[2])
        # HBM traffic of the dominant kernel: PMC counters of a separate rocprofv3 pass (scripts/gpu_visit4.sh `traffic`,
        # scripts/pmc_traffic.py), committed under profiles/; only used when it was measured for this very configuration
        traffic = traffic_detail = None
        try:
            import glob
            for fn in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")), reverse=True):
                tj = json.load(open(fn))
                def dims(cfg):
                    t_ = PATCH_CFGS.get(cfg[0])
                    return (t_[0] * t_[1], t_[2], cfg[1]) if t_ else (cfg[0], 0, cfg[1])
                if tj.get("cfg") and dims(tj["cfg"]) == dims(dom_tile) and tj.get("hbm_bytes_per_launch"):
                    esz = 2 if args.precision == "bf16" else 4
                    c0 = rb[0][1] if rb else None
                    alg = None if c0 is None else (c0["N"] * c0["H"] * c0["W"] * c0["cin"] * esz + c0["cout"] * c0["cin"] * 9 * esz
                                                   + c0["N"] * c0["OH"] * c0["OW"] * c0["cout"] * 4)
                    traffic = tj["hbm_bytes_per_launch"]
                    traffic_detail = {"hbm_bytes_per_launch": tj["hbm_bytes_per_launch"], "algorithmic_bytes_per_launch": alg,
                                      "source": "profiles/" + os.path.basename(fn) + " (separate rocprofv3 --pmc FETCH_SIZE / "
                                      "WRITE_SIZE passes, FETCH_SIZE x2 per the gfx950 correction; measured on the 1024->1024 "
                                      "3x3 layer)"}
                    break
        except Exception:
            traffic = traffic_detail = None
        roofline = {
            "bound": "mfma",
            "kernel": "%s<%s,%s> (tile config %d)" % (
                fam, "bf16" if args.precision == "bf16" else "f32", tile_name, dom_tile[0]),
            "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
            "traffic": traffic, "traffic_detail": traffic_detail,
            "avg_launch_us": round(a["ms"] * 1e3 / a["launches"], 2),
            "launches_per_frame": a["launches"] // nprof,
            "flop_per_launch": a["flops"] / a["launches"],
            "resblock_1024_tflops": None if rb_tf is None else round(rb_tf, 2),
            # the kernel figures above time every launch ALONE on the chip (eager single-stream replay); in the graph the
            # lanes share the chip, so the whole-frame rate is the utilisation actually reached in the timed region
            "frame_in_graph": {"achieved": round(sum(c["flops"] for c in fp.conv_log) / (elapsed / args.steps) / 1e12, 2),
                               "unit": "TFLOP/s", "frac": round(sum(c["flops"] for c in fp.conv_log) / (elapsed / args.steps) / 1e12 / peak, 4),
                               "note": "all conv FLOP of a frame / measured ms_per_step (norms, pooling, warp included in the time)"},
            "frame_ms_eager_events": round(sum(ms for _, _, ms in rows), 3),
            "per_kernel_ms": {k: round(v, 3) for k, v in sorted(total_ms.items(), key=lambda kv: -kv[1])},
        }
        if args.dump_ops:
            with open(args.dump_ops, "w") as f:
                tiles_by_label = {c["label"]: (c["tile"], c.get("splitk", 1), c.get("prefetch", 0)) for c in fp.conv_log}
                json.dump([dict(op=n_, label=l_, ms=ms, tile=tiles_by_label.get(l_) if n_ == KERNEL_FAMILY else None)
                           for n_, l_, ms in rows], f, indent=1)

    # ---------------- CPU baseline (oracle port on the host cores), rank 0 at N=1 only ----------
    cpu = None
    if rank == 0 and args.gpus == 1 and not args.no_cpu_baseline:
        from oracle import vid2vid_oracle as O
        # torch's default intra-op pool (one thread per physical core it detects); forcing
        # os.cpu_count() SMT threads measured 9x slower on the 2x64-core EPYC host
        ncores = torch.get_num_threads()
        sd = {k: v.detach().float().cpu() for k, v in model.netG0.state_dict().items()}
        fc = frames.cpu()
        if face:
            orc = O.InferenceOracle([sd], 0, False, False, [], opt.n_downsample_G, opt.n_blocks, opt.n_blocks_local)
            Ac = A.cpu()
            orc.step(Ac[:, 0:tG], fc[:, :tG - 1], None)                                             # warm-up frame
            c0 = time.perf_counter()
            for t in range(1, 1 + args.cpu_frames):
                orc.step(Ac[:, t:t + tG], None, None)
        else:
            orc = O.InferenceOracle([sd], 35, True, True, [26], opt.n_downsample_G, opt.n_blocks, opt.n_blocks_local)
            lc, ic = lab.cpu(), inst.cpu()
            orc.step(lc[0:tG].view(1, tG, 1, H, W), fc[:, :tG - 1], ic[0:tG].view(1, tG, 1, H, W))     # warm-up frame
            c0 = time.perf_counter()
            for t in range(1, 1 + args.cpu_frames):
                orc.step(lc[t:t + tG].view(1, tG, 1, H, W), None, ic[t:t + tG].view(1, tG, 1, H, W))
        cpu_s = time.perf_counter() - c0
        model_name = ""
        try:
            for line in open("/proc/cpuinfo"):
                if line.startswith("model name"):
                    model_name = line.split(":", 1)[1].strip(); break
        except OSError:
            pass
        cpu = {"value": round(args.cpu_frames / cpu_s, 4), "unit": "frames/s", "cores": ncores, "kind": "port",
               "sample": "%d frames (after 1 warm-up) of the same %dx%d workload, fp32, oracle/vid2vid_oracle.py on %s"
                         % (args.cpu_frames, W, H, model_name or "host CPU")}

    sys.stdout = _stdout
    if rank == 0:
        out = {
            "metric": "synthesized frames/sec (Vid2VidModelG.inference, %dx%d)" % (W, H),
            "value": round(fps, 3), "unit": "frames/s", "n_gpus": args.gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.precision, "data": "synthetic",
            "config": {"workload": "%s %dx%d inference, n_scales_spatial=%d, %s, ngf=128 n_blocks=9 "
                                   "(%.1fM params random-init, %.0f GFLOP/frame), batch 1 per sequence, 1 sequence per GPU"
                                   % (args.dataset, W, H, args.scales, "input_nc=15, no fg tower" if face else "--fg --use_instance", sum(q.numel() for q in model.parameters()) / 1e6,
                                      sum(c["flops"] for c in fp.conv_log) / 1e9),
                       "launches_per_frame": fp.plan.num_ops, "hipgraph": bool(opt.use_graph),
                       "graph_lanes": 3 if getattr(fp, "lanes", False) else 1,
                       "frame_tune": getattr(fp, "frame_tune_log", None),
                       "parallelism": "replicas x%d (independent sequences, no collective)" % args.gpus,
                       "output_finite": finite},
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        print(json.dumps(out))
        sys.stdout.flush()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
