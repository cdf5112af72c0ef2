"""The generator / discriminator rank roles (vid2vid_amd/roles.py) driven with the REAL model objects -- Vid2VidModelG,
Vid2VidModelD, FlowNet as create_model(opt) builds and wraps them -- on CPU with gloo and the backend in dry-run mode
(`networks.set_record_only`: every launch is argument-checked by the library, autograd graphs are built and walked,
optimizers step; nothing executes, so tensor VALUES are meaningless).

tests/test_cpu_roles.py proves the runtime's arithmetic with torch stand-ins (every loss / gradient / parameter equals a
single process).  What the stand-ins do not have, and this file exercises:
  * Vid2VidModelG.forward's `frame_range` / `first_chunk` path with per-scale fake_B_prev pyramids (n_scales_spatial = 2),
    the detach rules of niter_fix_global (coarse scale outside the optimizer), the encode_input / build_pyr plumbing;
  * flat-buffer optimizers (FusedAdam / FlatBuffers) under the per-role GradSync groups, NullOptimizer on the ranks that do
    not own a network;
  * the schedule calls the reference's update_models / init_params make on EVERY rank: update_fixed_params (both the
    reference-compatible default and the in-place rebuild), update_learning_rate, update_training_batch;
  * Vid2VidModelD.get_all_skipped_frames / the temporal scales through RoleFlowNet over several chunks of a sequence;
  * the start-up broadcast of everything outside the flat buffer (coarse-scale weights, norm running statistics).
"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

H, W = 64, 128


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, group, n_gen, fix, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from vid2vid_amd import networks as N, parallel, roles
    N.set_record_only(True)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from vid2vid_amd.options import make_opt
        from vid2vid_amd.models import create_model
        from vid2vid_amd.models.models import create_optimizer
        from vid2vid_amd.models.base_model import _UnsteppedAdam
        from vid2vid_amd.optim import FusedAdam
        torch.manual_seed(100 + rank)                     # DIFFERENT initial weights per rank: the start-up broadcast must align them
        opt = make_opt(isTrain=True, label_nc=35, use_instance=True, fg=True, ngf=8, ndf=8, n_blocks=2, n_blocks_local=1,
                       n_scales_spatial=2, n_downsample_G=2, loadSize=W, niter_fix_global=1, no_vgg=True, random_init_ok=True,
                       precision="fp32", gpu_ids=list(range(group)), n_gpus_gen=n_gen, n_frames_total=4, max_frames_per_gpu=1,
                       n_scales_temporal=1, num_D=1, niter=1, niter_decay=4, niter_step=1, fix_update_fixed_params=fix,
                       checkpoints_dir=os.environ["V2V_TEST_CKPT"], name="roles")
        models = create_model(opt)
        modelG, modelD, flowNet, optimizer_G, optimizer_D, optimizer_D_T = create_optimizer(opt, models)
        L = modelG.layout
        assert isinstance(modelG, roles.RoleModelG) and isinstance(modelD, roles.RoleModelD) and isinstance(flowNet, roles.RoleFlowNet)
        assert L.group_size == group and L.n_gen == n_gen and opt.gpu_ids == [rank] and opt.role_group_size == group
        mG, mD = modelG.module, modelD.module
        assert isinstance(optimizer_G, FusedAdam) == L.owns_G and isinstance(optimizer_D, FusedAdam) == L.owns_D
        assert isinstance(optimizer_D_T[0], FusedAdam) == L.owns_DT
        if L.owns_G:                                      # finest scale only (niter_fix_global): netG1 in the flat buffer, netG0 outside
            assert sum(p.numel() for p in optimizer_G.flat.params) == sum(p.numel() for p in mG.netG1.parameters())
            assert optimizer_G.grad_sync is not None and optimizer_G.grad_sync.group is L.pg_G
        # ---- start-up broadcast: every generator rank holds generator rank 0's netG0 / netG1 (weights AND running statistics)
        digest = torch.stack([torch.cat([v.detach().double().reshape(-1) for v in net.state_dict().values() if v.numel()]).sum()
                              for net in (mG.netG0, mG.netG1)])
        all_d = [torch.zeros_like(digest) for _ in range(world)]
        dist.all_gather(all_d, digest)
        for g in L.g_ranks:
            assert torch.equal(all_d[g], all_d[L.g_ranks[0]]), "generator ranks start from different weights"
        assert not torch.equal(all_d[L.d_ranks[0]], all_d[L.g_ranks[0]])       # the D-ranks' (unused) copies were drawn from another seed

        tG, tD, t_scales = opt.n_frames_G, opt.n_frames_D, opt.n_scales_temporal
        n_load = mG.n_frames_load
        assert n_load == n_gen and mG.n_frames_per_gpu == 1
        nT = opt.n_frames_total + tG - 1
        gen = torch.Generator().manual_seed(7 + L.seq)
        A = torch.randint(0, 35, (1, nT, 1, H, W), generator=gen).float()
        I = torch.randint(0, 5, (1, nT, 1, H, W), generator=gen).float()
        B = torch.tanh(torch.randn(1, nT, 3, H, W, generator=gen))
        reshape = lambda ts: [None if t is None else t.contiguous().view(-1, t.size(2), t.size(3), t.size(4)) for t in ts]
        log = {"chunks": 0, "T_active": 0, "shapes": None}

        def sequence():
            fake_B_prev_last, frames_all = None, (None, None, None, None)
            for i in range(0, opt.n_frames_total, n_load):
                sl = slice(i, i + n_load + tG - 1)
                fake_B, fake_B_raw, flow, weight, real_A, real_Bp, fake_B_last = modelG(A[:, sl], B[:, sl], I[:, sl], fake_B_prev_last)
                assert tuple(fake_B.shape) == (1, n_load, 3, H, W) and tuple(flow.shape) == (1, n_load, 2, H, W) and tuple(weight.shape) == (1, n_load, 1, H, W)
                assert tuple(real_A.shape) == (1, n_load, 36, H, W) and tuple(real_Bp.shape) == (1, n_load + 1, 3, H, W)
                if L.role == "G":                          # per-scale pyramid of the chunk's tail, detached (train.py:59-61)
                    assert [tuple(t.shape) for t in fake_B_last] == [(1, tG - 1, 3, H, W), (1, tG - 1, 3, H // 2, W // 2)]
                    assert not any(t.requires_grad for t in fake_B_last) and fake_B.requires_grad
                real_B_prev, real_B = real_Bp[:, :-1], real_Bp[:, 1:]
                flow_ref, conf_ref = flowNet(real_B, real_B_prev)
                fake_B_prev = mG.compute_fake_B_prev(real_B_prev, fake_B_prev_last, fake_B)
                fake_B_prev_last = fake_B_last
                losses = modelD(0, reshape([real_B, fake_B, fake_B_raw, real_A, real_B_prev, fake_B_prev, flow, weight, flow_ref, conf_ref]))
                assert len(losses) == len(mD.loss_names) and all(tuple(l.shape) == (1, 1) for l in losses)
                loss_dict = dict(zip(mD.loss_names, [torch.mean(x) for x in losses]))
                frames_all, skipped = mD.get_all_skipped_frames(frames_all, real_B, fake_B, flow_ref, conf_ref, t_scales, tD, n_load, i, flowNet)
                loss_dict_T = []
                for s in range(t_scales):
                    if skipped[0][s] is not None:
                        lt = modelD(s + 1, [f[s] for f in skipped])
                        loss_dict_T.append(dict(zip(mD.loss_names_T, [torch.mean(x) for x in lt])))
                        log["T_active"] += 1
                loss_G, loss_D, loss_D_T, t_act = mD.get_losses(loss_dict, loss_dict_T, t_scales)
                for loss, o in [(loss_G, optimizer_G), (loss_D, optimizer_D)] + [(loss_D_T[s], optimizer_D_T[s]) for s in range(t_act)]:
                    o.zero_grad(); loss.backward(); o.step()          # train.py:130-138 on EVERY rank
                log["chunks"] += 1

        sequence()                                         # epoch 1: the coarse scale is fixed
        step0 = optimizer_G.step_count if L.owns_G else None
        # ---- what update_models (models/models.py:139-160) calls at the end of epoch 1, on every rank ----
        mG.update_fixed_params()
        mG.update_learning_rate(2, "G"); mD.update_learning_rate(2, "D")
        mG.update_training_batch(1)
        assert mG.finetune_all and mG.n_frames_bp == 1
        if fix:
            assert mG.optimizer_G is optimizer_G           # same object (FusedAdam rebuilt in place, or the NullOptimizer)
            if L.owns_G:
                assert optimizer_G.step_count == 0 and optimizer_G.grad_sync.group is L.pg_G
                assert sum(p.numel() for p in optimizer_G.flat.params) == sum(p.numel() for n_ in (mG.netG0, mG.netG1) for p in n_.parameters())
        else:
            assert isinstance(mG.optimizer_G, _UnsteppedAdam) and mG._optimizer_G_live is optimizer_G
            if L.owns_G:
                assert optimizer_G.step_count == step0
        sequence()                                         # epoch 2 with the post-switch state; the captured handles keep stepping
        if L.owns_G:
            assert optimizer_G.step_count == (0 if fix else step0) + opt.n_frames_total // n_load
        mG.save("latest"); mD.save("latest")
        dist.barrier()
        q.put((rank, {"role": L.role, "chunks": log["chunks"], "T_active": log["T_active"], "owns": (L.owns_G, L.owns_D, L.owns_DT)}))
    except Exception:
        import traceback
        q.put((rank, {"error": traceback.format_exc()}))
    finally:
        parallel._ACTIVE_SYNCS.clear()
        dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.is_available(), reason="dry-run backend: a GPU-less host check")
@pytest.mark.parametrize("group,n_gen,fix", [(3, 2, False), (4, 2, True)])
def test_role_split_drives_the_real_models(tmp_path, group, n_gen, fix):
    """2 generator ranks + 1 (or 2) discriminator ranks run two sequences of the train.py chunk loop with the real
    Vid2VidModelG / Vid2VidModelD / FlowNet objects (2 spatial scales, niter_fix_global = 1, temporal scale 0) including
    the end-of-epoch schedule calls on every rank, without a deadlock, with the right shapes and optimizer ownership."""
    os.environ["V2V_TEST_CKPT"] = str(tmp_path)
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, group, port, group, n_gen, fix, q)) for r in range(group)]
    for p in procs:
        p.start()
    res = {}
    try:
        for _ in range(group):
            r, out = q.get(timeout=600)
            res[r] = out
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.terminate()
    for r, out in res.items():
        assert "error" not in out, "rank %d:\n%s" % (r, out.get("error"))
    for r, out in res.items():
        assert out["chunks"] == 2 * (4 // n_gen)
        assert out["T_active"] >= 1                         # the temporal discriminator ran once the history held tD frames
    files = sorted(os.listdir(os.path.join(str(tmp_path), "roles")))
    for f in ("latest_net_G0.pth", "latest_net_G1.pth", "latest_net_D.pth", "latest_net_D_T0.pth"):
        assert f in files, (f, files)
