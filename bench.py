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
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=2)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--profile-frames", type=int, default=3)
    ap.add_argument("--dump-ops", default="", help="write the per-op timing table (json) here")
    return ap.parse_args()


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
    lab, inst, frames = synthetic.label2city_sequence(L + tG, H, W, seed=1234 + rank, device=dev)
    A = lab.view(1, L + tG, 1, H, W)
    I = inst.view(1, L + tG, 1, H, W)

    def step(t):
        k = t % L
        model.inference(A[:, k:k + tG], frames[:, :tG - 1] if t == 0 else None, I[:, k:k + tG])

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
        # HBM traffic of the dominant kernel: PMC counters of a separate rocprofv3 pass (scripts/gpu_visit3.sh `traffic`,
        # scripts/pmc_traffic.py), committed under profiles/; only used when it was measured for this very configuration
        traffic = None
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
                    traffic = {"hbm_bytes_per_launch": tj["hbm_bytes_per_launch"], "algorithmic_bytes_per_launch": alg,
                               "source": "profiles/" + os.path.basename(fn) + " (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                               "passes, FETCH_SIZE x2 per the gfx950 correction; measured on the 1024->1024 3x3 layer)"}
                    break
        except Exception:
            traffic = None
        roofline = {
            "bound": "mfma",
            "kernel": "%s<%s,%s> (tile config %d)" % (
                fam, "bf16" if args.precision == "bf16" else "f32", tile_name, dom_tile[0]),
            "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
            "traffic": traffic,
            "avg_launch_us": round(a["ms"] * 1e3 / a["launches"], 2),
            "launches_per_frame": a["launches"] // nprof,
            "flop_per_launch": a["flops"] / a["launches"],
            "resblock_1024_tflops": None if rb_tf is None else round(rb_tf, 2),
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
        orc = O.InferenceOracle([sd], 35, True, True, [26], opt.n_downsample_G, opt.n_blocks, opt.n_blocks_local)
        lc, ic, fc = lab.cpu(), inst.cpu(), frames.cpu()
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
            "config": {"workload": "label2city %dx%d inference, n_scales_spatial=%d, --fg --use_instance, ngf=128 n_blocks=9 "
                                   "(%.1fM params random-init, %.0f GFLOP/frame), batch 1 per sequence, 1 sequence per GPU"
                                   % (W, H, args.scales, sum(q.numel() for q in model.parameters()) / 1e6,
                                      sum(c["flops"] for c in fp.conv_log) / 1e9),
                       "launches_per_frame": fp.plan.num_ops, "hipgraph": bool(opt.use_graph),
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
