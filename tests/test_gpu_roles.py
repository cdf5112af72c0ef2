"""The generator / discriminator rank roles with the REAL networks and REAL values.

One MI355X is all a test box has, and RCCL refuses several ranks on one device, so the three ranks of a sequence group
(2 generator ranks + 1 discriminator rank) share GPU 0 and talk over gloo with every transfer staged through host memory
(`parallel.host_staged`: the debugging transport; the production transport is RCCL device-to-device).  Everything else is
the product path: create_model(opt) -> roles.wrap_roles around Vid2VidModelG / Vid2VidModelD / FlowNet, the HIP kernels, the
flat-buffer FusedAdam under the per-role GradSync, the train.py chunk loop on every rank.

The yardstick is ONE process generating the same chunks itself (n_gpus_gen = 1, max_frames_per_gpu = 2): with n_frames_bp = 1
the previous frames are detached at every frame in both layouts (models/vid2vid_model_G.py:167-168), the discriminators
see the same batch of frames, so every loss, the all-reduced generator gradient, the D / D_T gradients and the parameters
after two chunks must agree to fp32 rounding.
"""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
H, W, N_CHUNKS = 64, 128, 2


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _opt(group, n_gen, ckpt):
    from vid2vid_amd.options import make_opt
    return make_opt(isTrain=True, label_nc=35, use_instance=True, fg=True, ngf=16, ndf=16, n_blocks=2, n_blocks_local=1,
                    n_scales_spatial=2, n_downsample_G=2, loadSize=W, niter_fix_global=0, no_vgg=True, random_init_ok=True,
                    precision="fp32", gpu_ids=list(range(group)), n_gpus_gen=n_gen, n_frames_total=4,
                    max_frames_per_gpu=2 // n_gen, n_scales_temporal=1, num_D=2, checkpoints_dir=ckpt, name="roles_gpu")


def _sequence(dev):
    gen = torch.Generator().manual_seed(11)
    nT = 4 + 2
    A = torch.randint(0, 35, (1, nT, 1, H // 8, W // 8), generator=gen).float().repeat_interleave(8, 3).repeat_interleave(8, 4)
    I = torch.randint(0, 6, (1, nT, 1, H // 8, W // 8), generator=gen).float().repeat_interleave(8, 3).repeat_interleave(8, 4)
    B = torch.tanh(torch.randn(1, nT, 3, H, W, generator=gen))
    return A.to(dev), I.to(dev), B.to(dev)


def _train(opt, modelG, modelD, flowNet, optimizer_G, optimizer_D, optimizer_D_T, dev):
    """train.py:47-93 for N_CHUNKS chunks of one sequence; returns the per-chunk record."""
    mG, mD = modelG.module, modelD.module
    with torch.no_grad():
        for si in range(opt.n_scales_spatial):
            getattr(mG, "netG%d" % si).model_final_flow[1].weight.mul_(0.1)
    tG, tD, t_scales, n_load = opt.n_frames_G, opt.n_frames_D, opt.n_scales_temporal, mG.n_frames_load
    assert n_load == 2
    A, I, B = _sequence(dev)
    reshape = lambda ts: [None if t is None else t.contiguous().view(-1, t.size(2), t.size(3), t.size(4)) for t in ts]
    fake_B_prev_last, frames_all, rec = None, (None, None, None, None), []
    for c in range(N_CHUNKS):
        i = c * n_load
        sl = slice(i, i + n_load + tG - 1)
        fake_B, fake_B_raw, flow, weight, real_A, real_Bp, fake_B_last = modelG(A[:, sl], B[:, sl], I[:, sl], fake_B_prev_last)
        real_B_prev, real_B = real_Bp[:, :-1], real_Bp[:, 1:]
        flow_ref, conf_ref = flowNet(real_B, real_B_prev)
        fake_B_prev = mG.compute_fake_B_prev(real_B_prev, fake_B_prev_last, fake_B)
        fake_B_prev_last = fake_B_last
        losses = modelD(0, reshape([real_B, fake_B, fake_B_raw, real_A, real_B_prev, fake_B_prev, flow, weight, flow_ref, conf_ref]))
        loss_dict = dict(zip(mD.loss_names, [torch.mean(x) for x in losses]))
        frames_all, skipped = mD.get_all_skipped_frames(frames_all, real_B, fake_B, flow_ref, conf_ref, t_scales, tD, n_load, i, flowNet)
        loss_dict_T = []
        for s in range(t_scales):
            if skipped[0][s] is not None:
                lt = modelD(s + 1, [f[s] for f in skipped])
                loss_dict_T.append(dict(zip(mD.loss_names_T, [torch.mean(x) for x in lt])))
        loss_G, loss_D, loss_D_T, t_act = mD.get_losses(loss_dict, loss_dict_T, t_scales)
        r = {"losses": {k: float(v.detach()) for k, v in list(loss_dict.items()) + [kv for d in loss_dict_T for kv in d.items()]},
             "fake_B": fake_B.detach().float().cpu()}
        for name, loss, o in [("G", loss_G, optimizer_G), ("D", loss_D, optimizer_D)] + [("DT", loss_D_T[s], optimizer_D_T[s]) for s in range(t_act)]:
            o.zero_grad(); loss.backward(); o.step()
            if hasattr(o, "flat"):
                torch.cuda.synchronize(dev)
                if o.grad_sync is not None:
                    o.grad_sync.wait_pending(dev)
                    torch.cuda.synchronize(dev)
                r["grad_" + name] = o.flat.flat_grad.detach().cpu().clone()
        rec.append(r)
    return rec


def _params(*opts):
    out = []
    for o in opts:
        out.append(o.flat.flat_param.detach().cpu().clone() if hasattr(o, "flat") else None)
    return out


def _worker(rank, world, port, group, n_gen, ckpt, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    try:
        import torch.distributed as dist
        from vid2vid_amd import parallel
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        if world > 1:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        from vid2vid_amd.models import create_model
        from vid2vid_amd.models.models import create_optimizer
        torch.manual_seed(0)                              # the same seeded networks in every process
        opt = _opt(group, n_gen, ckpt)
        models = create_model(opt)
        modelG, modelD, flowNet, optimizer_G, optimizer_D, optimizer_D_T = create_optimizer(opt, models)
        L = getattr(modelG, "layout", None)
        assert (L is not None) == (world > 1)
        rec = _train(opt, modelG, modelD, flowNet, optimizer_G, optimizer_D, optimizer_D_T, dev)
        torch.cuda.synchronize(dev)
        parallel.wait_pending()
        torch.cuda.synchronize(dev)
        pG, pD, pDT = _params(optimizer_G, optimizer_D, optimizer_D_T[0])
        # numpy arrays travel through the queue BY VALUE (torch tensors are handed over as shared-memory descriptors that die
        # with this process)
        np_ = lambda t: None if t is None else t.numpy()
        rec = [{k: (np_(v) if torch.is_tensor(v) else v) for k, v in r.items()} for r in rec]
        pG, pD, pDT = np_(pG), np_(pD), np_(pDT)
        out = {"rec": rec, "pG": pG, "pD": pD, "pDT": pDT,
               "role": None if L is None else L.role, "owns_D": True if L is None else L.owns_D, "owns_DT": True if L is None else L.owns_DT}
        if world > 1:
            dist.barrier()
        q.put((rank, out))
    except Exception:
        import traceback
        q.put((rank, {"error": traceback.format_exc()}))
    finally:
        try:
            import torch.distributed as dist
            if dist.is_initialized():
                dist.destroy_process_group()
        except Exception:
            pass


def _launch(world, group, n_gen, ckpt):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, group, n_gen, ckpt, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    try:
        for _ in range(world):
            r, out = q.get(timeout=900)
            res[r] = out
    finally:
        for p in procs:
            p.join(timeout=120)
            if p.is_alive():
                p.terminate()
    for r, out in res.items():
        assert "error" not in out, "rank %d:\n%s" % (r, out.get("error"))
    return res


def _close(a, b, what, tol):
    a, b = torch.from_numpy(a), torch.from_numpy(b)
    err = (a.double() - b.double()).abs().max().item() / max(b.double().abs().max().item(), 1e-12)
    print("%-60s max |d| / max |ref| = %.2e" % (what, err))
    assert err <= tol, "%s: relative difference %.3e > %.1e" % (what, err, tol)


def test_role_split_with_real_networks_equals_single_process(tmp_path):
    ref = _launch(1, 1, 1, str(tmp_path))[0]
    res = _launch(3, 3, 2, str(tmp_path))
    tol = 2e-4
    assert any("G_T_GAN" in k or "D_T_real" in k for k in ref["rec"][-1]["losses"]), "the temporal discriminator never ran"
    for r, out in res.items():
        for c in range(N_CHUNKS):
            got, want = out["rec"][c], ref["rec"][c]
            assert set(got["losses"]) == set(want["losses"]), (r, c, sorted(got["losses"]), sorted(want["losses"]))
            for name, v in want["losses"].items():
                assert abs(got["losses"][name] - v) <= 1e-3 * max(abs(v), 1e-3), (r, c, name, got["losses"][name], v)
            if out["role"] == "D":                          # the discriminator rank holds every frame the generator ranks sent it
                _close(got["fake_B"], want["fake_B"], "rank %d chunk %d fake_B" % (r, c), 1e-3)
            if out["role"] == "G":
                _close(got["grad_G"], want["grad_G"], "rank %d chunk %d all-reduced G gradient" % (r, c), tol)
            if out["owns_D"] and out["role"] == "D":
                _close(got["grad_D"], want["grad_D"], "rank %d chunk %d D gradient" % (r, c), tol)
                if "grad_DT" in want:
                    _close(got["grad_DT"], want["grad_DT"], "rank %d chunk %d D_T gradient" % (r, c), tol)
        if out["role"] == "G":
            _close(out["pG"], ref["pG"], "rank %d G parameters after %d chunks" % (r, N_CHUNKS), 5e-4)
        else:
            _close(out["pD"], ref["pD"], "rank %d D parameters" % r, 5e-4)
            _close(out["pDT"], ref["pDT"], "rank %d D_T parameters" % r, 5e-4)


def _bench_two_ranks(extra):
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--mode", "train", "--steps", "3", "--warmup", "1",
           "--no-train-parity", "--ngf", "32", "--width", "256", "--height", "128"] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads([ln for ln in r.stdout.splitlines() if ln.lstrip().startswith("{")][-1])


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two MI355X (the build environment's GPU boxes have one)")
def test_two_rank_rccl_data_parallel_bench_command():
    """the driver's N = 2 training command over REAL RCCL -- start-up broadcast, bucketed all-reduce of G / D / D_T
    gradients with G's buckets leaving from inside its backward pass, and the line's own account of how much collective time ran
    beside backward work (HIP event timestamps, parallel.GradSync.overlap_report)."""
    j = _bench_two_ranks([])
    c = j["config"]
    assert j["n_gpus"] == 2 and c["world_size"] == 2 and c["backend"] == "nccl" and c["output_finite"] is True
    assert c["grad_sync_buckets_inside_backward"] > 0 and c["grad_sync_allreduce_ms"] > 0 and c["grad_sync_hidden_ms"] >= 0


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two MI355X (the build environment's GPU boxes have one)")
def test_two_rank_rccl_role_mode_bench_command():
    """... and the n_gpus_gen split over real RCCL point-to-point: one generator rank + one discriminator rank share a sequence
    (models/models.py:10-23, vid2vid_model_G.py:126-133 of the reference; vid2vid_amd/roles.py)."""
    j = _bench_two_ranks(["--group-size", "2", "--n-gpus-gen", "1"])
    c = j["config"]
    assert j["n_gpus"] == 2 and c["world_size"] == 2 and c["backend"] == "nccl" and "role mode" in c["collective"]
    assert c["output_finite"] is True
