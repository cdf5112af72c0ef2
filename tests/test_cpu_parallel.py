"""Multi-process checks of the data-parallel runtime on CPU (gloo, world size 2): the flat-buffer
re-homing of parameters / gradients and the bucketed all-reduce that RCCL runs on the GPUs."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from vid2vid_amd.parallel import init_distributed, GradSync
    from vid2vid_amd.optim import FlatBuffers
    r, w, _ = init_distributed("gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(100 + rank)                       # ranks start from DIFFERENT weights
    net = nn.Sequential(nn.Conv2d(3, 5, 3), nn.BatchNorm2d(5), nn.Conv2d(5, 2, 1))
    before = [p.detach().clone() for p in net.parameters()]
    flat = FlatBuffers(list(net.parameters()))
    for p, b in zip(net.parameters(), before):          # re-homing preserves values, .grad views exist
        assert torch.equal(p.detach(), b)
        assert p.grad is not None and p.grad.data_ptr() >= flat.flat_grad.data_ptr()
    gs = GradSync(bucket_bytes=64)                       # tiny buckets -> many async all-reduces in flight
    assert len(gs.buckets(flat.flat_grad)) > 3
    gs.broadcast(flat.flat_param, src=0)                 # one start-up broadcast, no per-step replication
    ref0 = [torch.empty_like(b) for b in before]
    torch.manual_seed(100)
    net0 = nn.Sequential(nn.Conv2d(3, 5, 3), nn.BatchNorm2d(5), nn.Conv2d(5, 2, 1))
    for p, q0 in zip(net.parameters(), net0.parameters()):
        assert torch.equal(p.detach(), q0.detach())
    # each rank deposits rank-dependent gradients through the .grad views (as the HIP kernels do)
    for i, p in enumerate(net.parameters()):
        p.grad.fill_(float(rank + 1) * (i + 1))
    scale = gs.all_reduce(flat.flat_grad)
    assert scale == 1.0 / world
    for i, p in enumerate(net.parameters()):
        expect = sum(float(rk + 1) * (i + 1) for rk in range(world))
        assert torch.allclose(p.grad, torch.full_like(p.grad, expect))
        assert torch.allclose(p.grad * scale, torch.full_like(p.grad, expect / world))
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, "ok"))


def test_gradsync_and_flat_buffers_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    got = sorted(q.get(timeout=5) for _ in range(world))
    assert got == [(0, "ok"), (1, "ok")]


def test_single_process_is_a_noop():
    from vid2vid_amd.parallel import GradSync
    gs = GradSync()
    t = torch.arange(10.0)
    assert gs.all_reduce(t) == 1.0 and torch.equal(t, torch.arange(10.0))


def test_fused_adam_state_dict_roundtrip():
    """Optimizer-state checkpoint (extension over the reference, which saves none): moments, step count and
    hyper-parameters survive a save / load into a freshly built optimizer; a different network definition is refused."""
    import io
    import pytest
    import torch
    from vid2vid_amd.optim import FusedAdam
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 5, 3), torch.nn.Conv2d(5, 2, 1))
    opt = FusedAdam(list(net.parameters()), lr=1e-3, betas=(0.5, 0.999))
    opt.exp_avg.normal_()
    opt.exp_avg_sq.uniform_()
    opt.step_count = 17
    opt.param_groups[0]["lr"] = 2.5e-4
    buf = io.BytesIO()
    torch.save(opt.state_dict(), buf)
    buf.seek(0)
    net2 = torch.nn.Sequential(torch.nn.Conv2d(3, 5, 3), torch.nn.Conv2d(5, 2, 1))
    opt2 = FusedAdam(list(net2.parameters()), lr=1e-3, betas=(0.9, 0.999))
    opt2.load_state_dict(torch.load(buf))
    assert opt2.step_count == 17 and opt2.param_groups[0]["lr"] == 2.5e-4 and opt2.param_groups[0]["betas"] == (0.5, 0.999)
    assert torch.equal(opt2.exp_avg, opt.exp_avg) and torch.equal(opt2.exp_avg_sq, opt.exp_avg_sq)
    other = FusedAdam(list(torch.nn.Conv2d(3, 4, 3).parameters()))
    with pytest.raises(ValueError):
        other.load_state_dict(opt.state_dict())


def _tuning_worker(rank, world, port, q):
    import json
    import time
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    os.environ.pop("V2V_TUNE_CACHE", None)
    from vid2vid_amd.parallel import init_distributed, shared_tuning_cache
    init_distributed("gloo")
    release = shared_tuning_cache(rank, world)           # ranks > 0 block in here until rank 0 releases
    path = os.environ["V2V_TUNE_CACHE"]
    if rank == 0:
        assert not os.path.exists(path)                  # nothing measured yet
        time.sleep(0.5)                                  # "tile search"
        with open(path, "w") as f:
            json.dump({"1": {"1,2,3": [55, 1, 0]}}, f)
        release()
        seen = True
    else:
        seen = os.path.exists(path) and json.load(open(path))["1"]["1,2,3"] == [55, 1, 0]
        release()                                        # no-op
    dist.barrier()                                       # the timing barrier of bench.py: every rank gets here
    dist.destroy_process_group()
    q.put((rank, "ok" if seen else "rank %d did not see rank 0's selections" % rank))


def test_shared_tuning_cache_world2(tmp_path):
    """bench.py --gpus N: rank 0 searches tile configurations, the other ranks start only after its selections are on disk."""
    import tempfile
    port = _free_port()
    stale = os.path.join(tempfile.gettempdir(), "v2v_tune_%d.json" % port)
    open(stale, "w").write("{}")                         # must be removed by rank 0 before anybody reads it
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_tuning_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    if os.path.exists(stale):
        os.remove(stale)
    assert res == [(0, "ok"), (1, "ok")]


def test_update_fixed_params_default_is_the_reference_behaviour():
    """the reference's update_fixed_params stores a NEW Adam in self.optimizer_G (base_model.py:161-168)
    while train.py keeps stepping the one it captured (train.py:29) -- the finest scale trains on with its old moments, the
    coarse scales never move, update_learning_rate decays only the new object.  That is the default here."""
    import torch
    from vid2vid_amd import networks as N
    from vid2vid_amd.options import make_opt
    from vid2vid_amd.models.vid2vid_model_G import Vid2VidModelG
    from vid2vid_amd.models.base_model import _UnsteppedAdam
    if torch.cuda.is_available():
        pytest.skip("record-only construction is a CPU-host check")
    N.set_record_only(True)
    try:
        opt = make_opt(isTrain=True, label_nc=35, use_instance=True, fg=True, ngf=8, n_blocks=2, n_blocks_local=1,
                       n_scales_spatial=2, n_downsample_G=2, loadSize=64, niter_fix_global=3, no_vgg=True, random_init_ok=True,
                       precision="fp32", gpu_ids=[], n_gpus_gen=1, niter=10, niter_decay=10)
        G = Vid2VidModelG(); G.initialize(opt)
        captured = G.optimizer_G
        flat_ptr, n_fine = captured.flat.flat_param.data_ptr(), sum(p.numel() for p in G.netG1.parameters())
        captured.exp_avg.fill_(1.0); captured.step_count = 7
        lr0 = captured.param_groups[0]["lr"]
        G.update_fixed_params()
        # the captured optimizer is untouched: same flat buffer over the finest scale only, moments and step count kept
        assert G._optimizer_G_live is captured and captured.flat.flat_param.data_ptr() == flat_ptr
        assert sum(p.numel() for p in captured.flat.params) == n_fine and captured.step_count == 7 and float(captured.exp_avg.min()) == 1.0
        # self.optimizer_G is a fresh, never-stepped stand-in; finetune_all is set as in the reference
        assert isinstance(G.optimizer_G, _UnsteppedAdam) and G.optimizer_G is not captured and G.finetune_all
        assert not G._train_coarse                       # the unobservable coarse-scale gradients are not computed
        # update_learning_rate reaches the stand-in only (base_model.py:154-159 looks the attribute up again)
        G.update_learning_rate(15, "G")
        assert G.optimizer_G.param_groups[0]["lr"] == pytest.approx(opt.lr * 0.5) and captured.param_groups[0]["lr"] == lr0
        G.update_fixed_params()                          # idempotent (init_params and update_models may both call it)
        assert G._optimizer_G_live is captured
    finally:
        N.set_record_only(False)
        N._ENGINES.clear()


def test_update_fixed_params_rebuilds_the_captured_optimizer_in_place():
    """opt.fix_update_fixed_params: train.py captures optimizer_G once (train.py:29); with the fix selected
    update_fixed_params (base_model.py:162-168) must
    not leave that object stepping a flat buffer no parameter views.  After the call the SAME optimizer object owns every
    scale: each parameter's storage and gradient live inside its (new) flat buffers, values are preserved, moments are
    fresh, and the gradient synchroniser attached by parallel.sync_optimizers is still attached."""
    import torch
    from vid2vid_amd import networks as N
    from vid2vid_amd.options import make_opt
    from vid2vid_amd.models.vid2vid_model_G import Vid2VidModelG
    if torch.cuda.is_available():
        pytest.skip("record-only construction is a CPU-host check")
    N.set_record_only(True)
    try:
        opt = make_opt(isTrain=True, label_nc=35, use_instance=True, fg=True, ngf=8, n_blocks=2, n_blocks_local=1,
                       n_scales_spatial=2, n_downsample_G=2, loadSize=64, niter_fix_global=3, no_vgg=True, random_init_ok=True,
                       precision="fp32", gpu_ids=[], n_gpus_gen=1, fix_update_fixed_params=True)
        G = Vid2VidModelG(); G.initialize(opt)
        captured = G.optimizer_G                         # what create_optimizer() hands to train.py
        marker = object()
        captured.grad_sync = marker
        n_fine = sum(p.numel() for p in G.netG1.parameters())
        assert not G.finetune_all and sum(p.numel() for p in captured.flat.params) == n_fine
        def snapshot():
            return {"%d.%s" % (i, k): v.detach().clone() for i, net in enumerate((G.netG0, G.netG1)) for k, v in net.state_dict().items()}
        before = snapshot()
        old_flat = captured.flat.flat_param
        captured.exp_avg.fill_(1.0); captured.step_count = 7
        G.update_fixed_params()
        assert G.optimizer_G is captured and G.finetune_all and captured.grad_sync is marker
        flat, fgrad = captured.flat.flat_param, captured.flat.flat_grad
        assert flat.data_ptr() != old_flat.data_ptr()
        lo, hi = flat.data_ptr(), flat.data_ptr() + flat.numel() * 4
        glo, ghi = fgrad.data_ptr(), fgrad.data_ptr() + fgrad.numel() * 4
        params = list(G.netG0.parameters()) + list(G.netG1.parameters())
        assert len(captured.flat.params) == len(params)
        for p in params:
            assert lo <= p.data_ptr() < hi and glo <= p.grad.data_ptr() < ghi
        after = snapshot()
        for k, v in before.items():
            assert torch.equal(after[k], v), k
        assert captured.step_count == 0 and float(captured.exp_avg.abs().sum()) == 0.0
        assert captured.param_groups[0]["lr"] == G.old_lr and captured.param_groups[0]["betas"] == (opt.beta1, 0.999)
        # a write through the captured optimizer's flat buffer IS a write to the coarse-scale parameters
        flat.add_(1.0)
        assert torch.allclose(next(G.netG0.parameters()), before["0." + next(iter(G.netG0.state_dict()))] + 1.0)
    finally:
        N.set_record_only(False)
        N._ENGINES.clear()


def _bf16_wire_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from vid2vid_amd.parallel import init_distributed, GradSync
    init_distributed("gloo")
    torch.manual_seed(7)
    base = torch.randn(5000)
    grads = [base * (1.0 + 0.25 * r) + 0.01 * torch.randn(5000, generator=torch.Generator().manual_seed(r)) for r in range(world)]
    exact = sum(grads)
    flat = grads[rank].clone()
    gs = GradSync(bucket_bytes=4096, wire_dtype=torch.bfloat16)         # several buckets
    scale = gs.all_reduce(flat)
    rel = ((flat - exact).norm() / exact.norm()).item()
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, scale, rel, flat.dtype == torch.float32))


def test_bf16_wire_format_of_the_gradient_all_reduce():
    """GradSync(wire_dtype=torch.bfloat16) / V2V_GRAD_BF16=1: the buckets travel as bf16 (half the ring time on xGMI), the flat
    gradient stays fp32; the sum is the fp32 sum to bf16 rounding (< 1e-2 relative L2 here), every rank gets the same values."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bf16_wire_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=30)
    for rank, scale, rel, is_f32 in res:
        assert scale == 1.0 / world and is_f32 and rel < 1e-2, (rank, scale, rel)
    assert abs(res[0][2] - res[1][2]) < 1e-12
